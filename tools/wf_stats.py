import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
eps = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0**-21
ctx.set_retire_eps(eps)
sc = S.Mandelbulb(); schema = J.make_schema(sc, 3840, 2160, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
h = ctx.create_scene(sc); fb = ctx.create_framebuffer(3840, 2160); u = J.uniforms_from_schema(schema, (0.5, 1/3))
ctx.render_sample(h, fb, u, None, 1); ctx.debug_counters(True)
ctx.render_sample(h, fb, u, None, 1); c = ctx.debug_counters(True)
print("eps", eps)
for name, o in (("primary cheap", 0), ("primary full", 4), ("shadow cheap", 8), ("shadow full", 12)):
    rays, ls, ws = c[o], c[o+1], c[o+2]
    print(f"{name:14s} rays {rays:9d} lane-steps {ls:11d} wave-steps {ws:9d} steps/ray {ls/max(rays,1):6.1f} lanes/wave-step {ls/max(ws,1):5.1f}")
