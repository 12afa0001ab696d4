"""Headline frame, one sample at a time (RM_RENDER_NO_OVERLAP) and 3 in flight: python tools/time_alone.py a.so b.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        env = dict(os.environ, RM_LIB=os.path.abspath(lib) if lib != "default" else "")
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(lib, r.stdout.strip(), r.stderr.strip()[-200:])
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
out = []
for wl, sc, kw in (("c3b", S.Mandelbulb(), dict(width=3840, height=2160, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)),
                   ("c3a", S.Mandelbulb(), dict(width=3840, height=2160, counts=(256,), render_mode="preview", position=(0, 0, -2.5)))):
    schema = J.make_schema(sc, **kw); h = ctx.create_scene(sc); fb = ctx.create_framebuffer(kw["width"], kw["height"])
    u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
    for flags in (1 | abi.RM_RENDER_NO_OVERLAP, 1):
        ctx.render_timed(h, fb, u, 2, None, flags)
        out.append(min(ctx.render_timed(h, fb, u, 6, None, flags) for _ in range(3)))
    fb.destroy(); h.destroy()
print(f"c3b alone {out[0]:.2f} in-flight {out[1]:.2f} | c3a {out[2]:.2f} {out[3]:.2f}")
