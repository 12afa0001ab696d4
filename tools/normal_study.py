"""Image statistics of the fast build against the parity build of the same library (several scenes, 512x288, SPP samples)
and the headline timing: python tools/normal_study.py [lib.so ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        env = dict(os.environ, RM_LIB=os.path.abspath(lib) if lib != "default" else "")
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(lib); print(r.stdout.strip(), r.stderr.strip()[-300:])
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
N = int(os.environ.get("SPP", "32"))
noises = GC.halton_pairs(N)
CASES = [("bulb [256] 1 light", S.Mandelbulb(), dict(counts=(256,), position=(0, 0, -2.5), lights=GC.LIGHT)),
         ("bulb [128,64] soft", S.Mandelbulb(), dict(counts=(128, 64), position=(0, 0, -2.5), lights=GC.SOFT_LIGHT)),
         ("csg64 [128] 1 light", S.csg64(), dict(counts=(128,), position=(0, 0, -5.0), lights=GC.LIGHT)),
         ("fractal1 live", S.SphereGridFractal(), dict(counts=(128, 128, 64, 32, 32), position=(0, 0, 0))),
         ("menger [128,64]", S.MengerSponge(), dict(counts=(128, 64), position=(0, 0, -3.0), lights=GC.LIGHT)),
         ("kifs tree [128,64]", S.KifsTree(), dict(counts=(128, 64), position=(0, 0, -3.0), lights=GC.LIGHT))]
for name, sc, kw in CASES:
    h = ctx.create_scene(sc)
    def render(flags):
        schema = J.make_schema(sc, 512, 288, render_mode="full", **kw)
        fb = ctx.create_framebuffer(512, 288)
        for n in noises: ctx.render_sample(h, fb, J.uniforms_from_schema(schema, n), None, flags)
        out = fb.download(0); fb.destroy(); return out[..., :3] / N
    ref, got = render(0), render(1)
    fin = np.isfinite(ref).all(-1) & np.isfinite(got).all(-1)
    d = np.abs(got - ref); m = (d.max(-1) > 0) & fin
    # standard error of the ratio from the per-pixel differences
    diff = (got[m] - ref[m]).mean(-1)
    se = diff.std() / np.sqrt(max(diff.size, 1)) / ref[m].mean()
    print(f"  {name:22s} differing px {m.mean():.3f}: fast/strict {got[m].mean()/ref[m].mean():.4f} +- {se:.4f}; all px {got[fin].mean()/ref[fin].mean():.4f}; rmse {np.sqrt((d[fin]**2).mean()):.5f}")
    h.destroy()
sc = S.Mandelbulb(); h = ctx.create_scene(sc)
big = J.make_schema(sc, 3840, 2160, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
fbb = ctx.create_framebuffer(3840, 2160); ub = J.uniforms_from_schema(big, (0.5, 1/3))
ctx.render_timed(h, fbb, ub, 1, None, 1); ms = min(ctx.render_timed(h, fbb, ub, 3, None, 1) for _ in range(3))
print(f"  C3b {ms:.2f} ms")
