"""Lane-refilling kernel (RM_RENDER_STREAM) against the one-thread-one-pixel kernel: same bits, and the headline timing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_cases as GC
import test_gpu_parity as T
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
ST, NO = abi.RM_RENDER_STREAM, abi.RM_RENDER_NO_OVERLAP
if "--no-check" not in sys.argv:
    for case in ("sphere_full", "sphere_full_light", "sphere_full_3b_soft_4spp", "sphere_full_dof_fog", "sphere_full_mix_2spp", "csg64_full_light",
                 "csg_mixed_full_2b", "lattice_full_2b", "fractal1_full_2b", "mandelbulb_full_light"):
        sc, samples, schema = GC.image_schema(case)
        noises = T.load("image_" + case)["rand_noise"]
        for build in (T.STRICT, T.FAST):
            a = T.render_gpu(ctx, sc, schema, noises, build | T.MK | NO)
            b = T.render_gpu(ctx, sc, schema, noises, build | ST | NO)
            c = T.render_gpu(ctx, sc, schema, noises, build | ST)
            bad = [int((~T.same_bits(a[k], b[k])).sum()) for k in range(3)] + [int((~T.same_bits(a[k], c[k])).sum()) for k in range(3)]
            print(case, "strict" if build == T.STRICT else "fast", "differing elements", bad)
W = [("c3b", S.Mandelbulb(), dict(width=3840, height=2160, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)),
     ("live", S.SphereGridFractal(), dict(width=1280, height=720, counts=(128, 128, 64, 32, 32), render_mode="full", position=(0, 0, 0))),
     ("menger", S.MengerSponge(), dict(width=1920, height=1080, counts=(128, 64), render_mode="full", position=(0, 0, -3.0), lights=GC.LIGHT)),
     ("c4/8", S.csg64(), dict(width=4096, height=512, counts=(128,), render_mode="full", position=(0, 0, -5.0), lights=GC.LIGHT))]
for wl, sc, kw in W:
    schema = J.make_schema(sc, **kw); h = ctx.create_scene(sc); fb = ctx.create_framebuffer(kw["width"], kw["height"])
    u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
    row = []
    for flags in (1 | T.MK | NO, 1 | ST | NO, 1 | T.MK, 1 | ST):
        ctx.render_timed(h, fb, u, 1, None, flags)
        row.append(min(ctx.render_timed(h, fb, u, 4, None, flags) for _ in range(3)))
    print(f"{wl}: pixel kernel alone {row[0]:.2f}  stream alone {row[1]:.2f} | 3 in flight: pixel {row[2]:.2f} stream {row[3]:.2f}")
    fb.destroy(); h.destroy()
