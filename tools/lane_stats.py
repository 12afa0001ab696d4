"""Lane use inside the Mandelbulb evaluation on the headline frame (diagnostic build -DRM_LANE_STATS):
RM_LIB=gpurun_in/lib_stats.so python tools/lane_stats.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
lib = native.load_library()
sc = S.Mandelbulb(); kw = dict(width=3840, height=2160, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
schema = J.make_schema(sc, **kw); h = ctx.create_scene(sc); fb = ctx.create_framebuffer(kw["width"], kw["height"])
u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
out = (ctypes.c_ulonglong * 4)()
for name, flags in (("megakernel", 1 | 4),):  # (the wavefront pipeline is in the tests' cross-check build only since round 5)
    lib.rm_fast_lane_stats(out, 1)
    ctx.render_timed(h, fb, u, 1, None, flags)
    lib.rm_fast_lane_stats(out, 1)
    lr, wr, la, wa = [int(x) for x in out]
    px = kw["width"] * kw["height"]
    print(f"{name}: evaluations/pixel {la/px:.1f}; lanes active per issued evaluation {la/wa:.3f}; "
          f"rounds used / issued {lr/wr:.3f}; mean rounds per evaluation {lr/la:.2f}; issued lane-rounds per pixel {wr/px:.0f}")
