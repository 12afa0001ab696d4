"""Time library variants on the headline config: python tools/time_variants.py a.so b.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 or (len(sys.argv) == 2 and sys.argv[1] != "--child"):
    for lib in sys.argv[1:]:
        env = dict(os.environ, RM_LIB=os.path.abspath(lib) if lib != "default" else "")
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(lib, r.stdout.strip().replace("\n", " | "), r.stderr.strip()[-300:])
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
for wl, sc, kw in (("c3b", S.Mandelbulb(), dict(width=3840, height=2160, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)),
                   ("c3a", S.Mandelbulb(), dict(width=3840, height=2160, counts=(256,), render_mode="preview", position=(0, 0, -2.5))),
                   ("live", S.SphereGridFractal(), dict(width=1280, height=720, counts=(128, 128, 64, 32, 32), render_mode="full", position=(0, 0, 0))),
                   ("menger", S.MengerSponge(), dict(width=1920, height=1080, counts=(128, 64), render_mode="full", position=(0, 0, -3.0), lights=GC.LIGHT)),
                   ("kifs", S.KifsTree(), dict(width=1920, height=1080, counts=(128, 64), render_mode="full", position=(0, 0, -3.0), lights=GC.LIGHT)),
                   ("lattice", S.SphereLattice(), dict(width=1920, height=1080, counts=(128, 64), render_mode="full", position=(0, 0, -3.0), lights=GC.LIGHT)),
                   ("kbox", S.KifsBox(), dict(width=1920, height=1080, counts=(128, 64), render_mode="full", position=(0, 0, -3.0), lights=GC.LIGHT)),
                   ("c4/8", S.csg64(), dict(width=4096, height=512, counts=(128,), render_mode="full", position=(0, 0, -5.0), lights=GC.LIGHT))):
    schema = J.make_schema(sc, **kw); h = ctx.create_scene(sc); fb = ctx.create_framebuffer(kw["width"], kw["height"])
    u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
    ctx.render_timed(h, fb, u, 1, None, 1)
    ms = min(ctx.render_timed(h, fb, u, 3, None, 1) for _ in range(3))
    print(f"{wl} {ms:.2f}")
    fb.destroy(); h.destroy()
