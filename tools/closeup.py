"""A frame in which (almost) every pixel hits the Mandelbulb: all four waves of every workgroup stay busy.
Used under rocprofv3 --pmc to read the VALU issue rate without the sky/silhouette imbalance of the headline frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
sc = S.Mandelbulb(); h = ctx.create_scene(sc)
W, H = 3840, 2160
schema = J.make_schema(sc, W, H, counts=(256,), render_mode="full", position=(0, 0, float(os.environ.get("CAMZ", "-2.5"))), lights=GC.LIGHT, fov=float(os.environ.get("FOV", "0.35")))
fb = ctx.create_framebuffer(W, H)
u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
ctx.render_timed(h, fb, u, 1, None, 1)
ms = min(ctx.render_timed(h, fb, u, 2, None, 1) for _ in range(2))
print(f"closeup {ms:.2f} ms")
