"""Headline frame (3840x2160 Mandelbulb, full, [256], the point light): ms per sample of the fast build, the strict build and the
strict build in the GL stack's arithmetic (rm_ctx_set_gl_stack)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raymarching_engine_amd import abi, job as J, native, scene as S

ctx = native.Context(0)
sc = S.Mandelbulb()
schema = J.make_schema(sc, 3840, 2160, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=[J.point_light((2.0, 3.0, -4.0))])
u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
h = ctx.create_scene(sc)
fb = ctx.create_framebuffer(3840, 2160)
for name, gl, flags in (("fast", 0, abi.RM_RENDER_FAST), ("strict", 0, abi.RM_RENDER_STRICT | abi.RM_RENDER_MEGAKERNEL), ("strict, GL stack's arithmetic", 1, abi.RM_RENDER_STRICT | abi.RM_RENDER_MEGAKERNEL)):
    ctx.set_gl_stack(gl)
    ctx.render_timed(h, fb, u, 2, None, flags)
    print(f"{name}: {ctx.render_timed(h, fb, u, 5, None, flags):.2f} ms per sample (rm_render_timed: the mean of 5)")
