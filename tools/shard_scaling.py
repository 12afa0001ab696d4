"""Kernel time of every rank's stripe set when the headline frame is dealt over N GPUs (all measured on one GPU):
the compute part of strong scaling = full-frame time / max over ranks."""
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
SR = int(sys.argv[1]) if len(sys.argv) > 1 else shard.STRIPE_ROWS
print("stripe rows", SR)
full = None
for n in (1, 4, 8):
    times = []
    for r in range(n):
        fb = ctx.create_striped_framebuffer(W, H, SR, n, r)
        ctx.render_timed(h, fb, u, 1, None, 1)
        times.append(min(ctx.render_timed(h, fb, u, 3, None, 1) for _ in range(2)))
        fb.destroy()
    if n == 1: full = times[0]
    print(f"N={n}: per-rank kernel ms min {min(times):.3f} max {max(times):.3f}; compute speed-up {full/max(times):.2f}x of {n}")
