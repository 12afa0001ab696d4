import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_cases as GC
from oracle import oracle as O
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0); O.set_tan_mode(O.TAN_PORTABLE)
def gpu(sc, schema, noises, flags):
    r = schema["render"]; h = ctx.create_scene(sc); fb = ctx.create_framebuffer(r["width"], r["height"])
    for n in noises: ctx.render_sample(h, fb, J.uniforms_from_schema(schema, tuple(n)), None, flags)
    out = [fb.download(p) for p in range(3)]; fb.destroy(); h.destroy(); return out
def orc(sc, schema, noises):
    r = schema["render"]; fr = O.Frame(r["width"], r["height"])
    for n in noises: O.render(sc, J.uniforms_from_schema(schema, tuple(n)), fr, threads=O.host_cores())
    return [fr.color, fr.normal_dof, fr.albedo_depth]
def rel(a, b):
    with np.errstate(invalid="ignore"): d = np.abs(a - b) / np.maximum(1.0, np.abs(a))
    d[(a == b) | (np.isnan(a) & np.isnan(b))] = 0; d[np.isnan(d)] = np.inf
    return d.max(-1)
sc = S.single_sphere()
for name, kw, ns in [
    ("3b nolight 1spp", dict(counts=(64,32,32)), 1),
    ("2b nolight 1spp", dict(counts=(64,32)), 1),
    ("2b 64,64 nolight", dict(counts=(64,64)), 1),
    ("1b soft", dict(counts=(64,), lights=GC.SOFT_LIGHT), 1),
    ("2b point", dict(counts=(64,32), lights=GC.LIGHT), 1),
    ("3b soft 1spp", dict(counts=(64,32,32), lights=GC.SOFT_LIGHT), 1),
    ("3b soft 4spp", dict(counts=(64,32,32), lights=GC.SOFT_LIGHT), 4),
]:
    schema = J.make_schema(sc, 64, 32, render_mode="full", **kw)
    noises = GC.halton_pairs(ns)
    want = orc(sc, schema, noises); s = gpu(sc, schema, noises, 0); f = gpu(sc, schema, noises, 1)
    for nm, g in (("strict", s), ("fast", f)):
        d = rel(want[0], g[0]); fin = np.isfinite(want[0]).all(-1) & np.isfinite(g[0]).all(-1)
        print(f"{name:20s} {nm:6s} >1e-5 {np.mean(d>1e-5):.4f} >1e-2 {np.mean(d>1e-2):.4f} nan ref {np.isnan(want[0]).any(-1).mean():.3f} got {np.isnan(g[0]).any(-1).mean():.3f} mean ref {want[0][fin][:,:3].mean():.5f} got {g[0][fin][:,:3].mean():.5f}")
    d = rel(want[0], f[0]); bad = np.argwhere(d > 1e-2)[:3]
    for y, x in bad: print("   px", y, x, "ref", want[0][y,x], "fast", f[0][y,x], "strict", s[0][y,x])
