"""C5's stripes with other step budgets: are the 64-step bounces dear because their escaping shadow rays cannot be jumped (72 steps needed)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raymarching_engine_amd import abi, job as J, native, scene as S, shard
ctx = native.Context(0)
F, MK = abi.RM_RENDER_FAST, abi.RM_RENDER_MEGAKERNEL
soft = [J.point_light((2.0, 3.0, -4.0), size=0.3)]
sc = S.csg64(); h = ctx.create_scene(sc)
fb = ctx.create_striped_framebuffer(8192, 8192, shard.STRIPE_ROWS, 8, 0)
for counts in ((128, 64, 64), (128, 128, 128), (128, 96, 96), (128, 80, 80), (128, 72, 72), (128, 56, 56), (128, 48, 48), (128, 32, 32), (64, 64, 64), (128,), (96,), (72,), (64,), (128, 64), (128, 128)):
    schema = J.make_schema(sc, 8192, 8192, counts=counts, render_mode="full", position=(0, 0, -5.0), lights=soft)
    u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
    ctx.render_timed(h, fb, u, 1, None, F | MK | abi.RM_RENDER_NO_OVERLAP)
    ms = min(ctx.render_timed(h, fb, u, 2, None, F | MK | abi.RM_RENDER_NO_OVERLAP) for _ in range(2))
    print(counts, f"{ms:.2f} ms")
