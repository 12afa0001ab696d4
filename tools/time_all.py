"""Timing table on one MI355X: workloads x (fast|strict) x (wavefront|megakernel).  ms per sample (HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
xctx = native.Context(0, library=native.XCHECK_LIB_PATH)  # the wavefront pipeline: the tests' cross-check build of the same sources
W = [("C2 sphere 1920x1080 preview [128]", S.single_sphere(), dict(width=1920, height=1080, counts=(128,), render_mode="preview")),
     ("C3a bulb 3840x2160 preview [256]", S.Mandelbulb(), dict(width=3840, height=2160, counts=(256,), render_mode="preview", position=(0, 0, -2.5))),
     ("C3b bulb 3840x2160 full [256] 1 light", S.Mandelbulb(), dict(width=3840, height=2160, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)),
     ("C4 csg64 4096x512 shard full [128] 1 light", S.csg64(), dict(width=4096, height=512, counts=(128,), render_mode="full", position=(0, 0, -5.0), lights=GC.LIGHT)),
     ("C4 csg64 4096x4096 full [128] 1 light", S.csg64(), dict(width=4096, height=4096, counts=(128,), render_mode="full", position=(0, 0, -5.0), lights=GC.LIGHT)),
     ("C5 csg64 8192x1024 shard full [128,64,64] soft", S.csg64(), dict(width=8192, height=1024, counts=(128, 64, 64), render_mode="full", position=(0, 0, -5.0), lights=GC.SOFT_LIGHT)),
     ("default live job 1280x720 full [128,128,64,32,32] fractal1", S.SphereGridFractal(), dict(width=1280, height=720, counts=(128, 128, 64, 32, 32), render_mode="full", position=(0, 0, 0)))]
print(f"{'workload':58s} {'fast wf':>10s} {'fast mk':>10s} {'strict wf':>10s} {'strict mk':>10s}   Mpix/s (fast wf)")
for name, sc, kw in W:
    schema = J.make_schema(sc, **kw)
    hs = {id(c): c.create_scene(sc) for c in (ctx, xctx)}; fbs = {id(c): c.create_framebuffer(kw["width"], kw["height"]) for c in (ctx, xctx)}
    u = J.uniforms_from_schema(schema, (0.5, 1 / 3)); row = []
    for flags in (1 | 16 | 32, 1 | 4 | 32, 0 | 16 | 32, 0 | 4 | 32):  # fast/strict x forced wavefront/pixel kernel, one sample at a time
        heavy = "csg64" in name and not (flags & 1)
        c = xctx if flags & 16 else ctx
        c.render_timed(hs[id(c)], fbs[id(c)], u, 1, None, flags)
        row.append(min(c.render_timed(hs[id(c)], fbs[id(c)], u, 1 if heavy else 3, None, flags) for _ in range(1 if heavy else 2)))
    print(f"{name:58s} " + " ".join(f"{x:10.2f}" for x in row) + f"   {kw['width']*kw['height']/row[0]/1e3:8.0f}")
    for c in (ctx, xctx):
        fbs[id(c)].destroy(); hs[id(c)].destroy()
