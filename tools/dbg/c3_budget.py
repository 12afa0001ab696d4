"""The headline scene under other step budgets and bounce counts: does any budget cost more than a larger one (rays that escape but cannot be jumped)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
F, MK = abi.RM_RENDER_FAST, abi.RM_RENDER_MEGAKERNEL
sc = S.Mandelbulb(); h = ctx.create_scene(sc)
fb = ctx.create_framebuffer(3840, 2160)
for counts in ((256,), (128,), (64,), (48,), (32,), (24,), (16,), (256, 128), (256, 64), (256, 32), (256, 16), (128, 128, 64, 32, 32)):
    for lights in (GC.LIGHT, []):
        schema = J.make_schema(sc, 3840, 2160, counts=counts, render_mode="full", position=(0, 0, -2.5), lights=lights)
        u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
        ctx.render_timed(h, fb, u, 1, None, F | MK | abi.RM_RENDER_NO_OVERLAP)
        ms = min(ctx.render_timed(h, fb, u, 2, None, F | MK | abi.RM_RENDER_NO_OVERLAP) for _ in range(2))
        print(counts, "light" if lights else "no light", f"{ms:.3f} ms")
