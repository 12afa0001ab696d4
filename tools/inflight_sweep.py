"""Samples in flight: ms per sample of the headline frame (whole, and rank 0's stripes of 2/4/8) against the depth."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S, shard
ctx = native.Context(0)
sc = S.Mandelbulb(); h = ctx.create_scene(sc)
W, H = 3840, 2160
schema = J.make_schema(sc, W, H, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
print("parts  alone " + " ".join(f"depth{d:>2d}" for d in (1, 2, 3, 4, 6, 8)))
for n in (1, 2, 4, 8):
    fb = ctx.create_striped_framebuffer(W, H, shard.STRIPE_ROWS, n, 0)
    row = []
    ctx.render_timed(h, fb, u, 2, None, 1 | abi.RM_RENDER_NO_OVERLAP)
    row.append(min(ctx.render_timed(h, fb, u, 8, None, 1 | abi.RM_RENDER_NO_OVERLAP) for _ in range(2)))
    for d in (1, 2, 3, 4, 6, 8):
        ctx.set_samples_in_flight(d)
        ctx.render_timed(h, fb, u, 8, None, 1)
        row.append(min(ctx.render_timed(h, fb, u, 16, None, 1) for _ in range(2)))
    print(f"{n:5d} " + " ".join(f"{x:7.3f}" for x in row))
    fb.destroy()
