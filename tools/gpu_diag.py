"""Diagnostics on a GPU box: error distributions of the strict/fast builds against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_cases as GC
from oracle import oracle as O
from raymarching_engine_amd import abi, job as J, native, scene as S

ctx = native.Context(0)
O.set_tan_mode(O.TAN_PORTABLE)

def gpu(sc, schema, noises, flags):
    r = schema["render"]; h = ctx.create_scene(sc); fb = ctx.create_framebuffer(r["width"], r["height"])
    for n in noises: ctx.render_sample(h, fb, J.uniforms_from_schema(schema, tuple(n)), None, flags)
    out = [fb.download(p) for p in range(3)]; fb.destroy(); h.destroy(); return out

def orc(sc, schema, noises):
    r = schema["render"]; fr = O.Frame(r["width"], r["height"])
    for n in noises: O.render(sc, J.uniforms_from_schema(schema, tuple(n)), fr, threads=O.host_cores())
    return [fr.color, fr.normal_dof, fr.albedo_depth]

def rel(a, b):
    with np.errstate(invalid="ignore"):
        d = np.abs(a - b) / np.maximum(1.0, np.abs(a))
    d[(a == b) | (np.isnan(a) & np.isnan(b))] = 0; d[np.isnan(d)] = np.inf
    return d.max(-1)

cases = list(GC.IMAGES) 
for case in cases:
    sc, samples, schema = GC.image_schema(case)
    noises = GC.halton_pairs(samples)
    want = orc(sc, schema, noises)
    for name, flags in (("strict", 0), ("fast", 1)):
        got = gpu(sc, schema, noises, flags)
        d = rel(want[0], got[0])
        fin = np.isfinite(want[0]).all(-1) & np.isfinite(got[0]).all(-1)
        m = lambda t: float(np.mean(d > t))
        print(f"{case:28s} {name:6s} >1e-5 {m(1e-5):.4f} >1e-3 {m(1e-3):.4f} >1e-2 {m(1e-2):.4f} >5e-2 {m(5e-2):.4f} >0.2 {m(0.2):.4f}  mean ref {want[0][fin][:,:3].mean():.5f} got {got[0][fin][:,:3].mean():.5f}")
# bigger lit mandelbulb
sc = S.Mandelbulb()
schema = J.make_schema(sc, 256, 128, counts=(256,), render_mode="full", position=(0,0,-2.5), lights=GC.LIGHT)
noises = GC.halton_pairs(1)
want = orc(sc, schema, noises)
for name, flags in (("strict", 0), ("fast", 1)):
    got = gpu(sc, schema, noises, flags)
    d = rel(want[0], got[0]); m = lambda t: float(np.mean(d > t))
    print(f"bulb256 [256] lit           {name:6s} >1e-5 {m(1e-5):.4f} >1e-3 {m(1e-3):.4f} >1e-2 {m(1e-2):.4f} >5e-2 {m(5e-2):.4f} >0.2 {m(0.2):.4f} mean ref {want[0][...,:3].mean():.5f} got {got[0][...,:3].mean():.5f}")
# 16 spp means
fr = O.Frame(256,128)
n16 = GC.halton_pairs(16)
want = orc(sc, schema, n16)
for name, flags in (("strict", 0), ("fast", 1)):
    got = gpu(sc, schema, n16, flags)
    d = rel(want[0]/16, got[0]/16); m = lambda t: float(np.mean(d > t))
    print(f"bulb256 16spp mean          {name:6s} >1e-3 {m(1e-3):.4f} >1e-2 {m(1e-2):.4f} >5e-2 {m(5e-2):.4f} rmse {np.sqrt(np.mean((want[0][...,:3]-got[0][...,:3])**2))/16:.5f}")
# speed of both builds on the headline config
sc3 = S.Mandelbulb(); sch3 = J.make_schema(sc3, 3840, 2160, counts=(256,), render_mode="full", position=(0,0,-2.5), lights=GC.LIGHT)
h = ctx.create_scene(sc3); fb = ctx.create_framebuffer(3840, 2160); u = J.uniforms_from_schema(sch3, (0.5, 1/3))
for name, flags in (("fast", 1), ("strict", 0)):
    ctx.render_timed(h, fb, u, 1, None, flags)
    ms = ctx.render_timed(h, fb, u, 3, None, flags)
    print(f"C3b {name}: {ms:.2f} ms/launch = {3840*2160/ms/1e3:.1f} Mpix/s")
