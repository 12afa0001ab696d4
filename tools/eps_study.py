"""Effect of the retire tolerance on speed and on the image (fast build vs strict build, Mandelbulb lit)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
sc = S.Mandelbulb(); h = ctx.create_scene(sc)
def render(w, hgt, noises, flags):
    schema = J.make_schema(sc, w, hgt, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
    fb = ctx.create_framebuffer(w, hgt)
    for n in noises: ctx.render_sample(h, fb, J.uniforms_from_schema(schema, n), None, flags)
    out = fb.download(0); fb.destroy(); return out
NS = int(os.environ.get("SPP", "16")); noises = GC.halton_pairs(NS)
ref = render(512, 288, noises, 0)[..., :3] / NS
big = J.make_schema(sc, 3840, 2160, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
fbb = ctx.create_framebuffer(3840, 2160); ub = J.uniforms_from_schema(big, (0.5, 1/3))
for eps in [float(x) for x in os.environ.get("EPS_LIST", "0,4.76837158203125e-07,2e-6,1e-5,1e-4").split(",")]:
    ctx.set_retire_eps(eps)
    got = render(512, 288, noises, 1)[..., :3] / NS
    hit = ref.std(-1) < 1e9
    d = np.abs(got - ref)
    ctx.render_timed(h, fbb, ub, 1, None, 1); ms = min(ctx.render_timed(h, fbb, ub, 3, None, 1) for _ in range(2))
    m = d.max(-1) > 0
    print(f"lit {m.mean():.3f} ratio {got[m].mean()/ref[m].mean():.4f}", end=" ")
    print(f"eps {eps:9.3g}  C3b {ms:6.2f} ms {3840*2160/ms/1e3:6.0f} Mpix/s | 16spp 512x288 vs strict: mean {got.mean():.5f} (strict {ref.mean():.5f}) rmse {np.sqrt((d**2).mean()):.5f} p99 abs {np.percentile(d,99):.4f}")
