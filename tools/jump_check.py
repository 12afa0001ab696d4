#!/usr/bin/env python3
"""Round 3: the far-field jump of the fast Mandelbulb march (rm_device.hpp far_jump).
(1) the planes with the jump against RM_RENDER_NO_FAR_JUMP on the headline frame, 2 samples, both implementations: bit for bit;
(2) kernel time of both, headline (C3b) and preview (C3a), and of the other scene kinds (regressions).
    python tools/jump_check.py [lib.so ...]   ("default" = the product library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) >= 2 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        env = dict(os.environ, RM_LIB=os.path.abspath(lib) if lib != "default" else "")
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print("==", lib); print(r.stdout.strip()); print(r.stderr.strip()[-600:])
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
F, NJ, NO = abi.RM_RENDER_FAST, abi.RM_RENDER_NO_FAR_JUMP, abi.RM_RENDER_NO_OVERLAP
sc = S.Mandelbulb()
W, H = 3840, 2160
schema = J.make_schema(sc, W, H, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
h = ctx.create_scene(sc)
noises = GC.halton_pairs(2)
planes = {}
xctx = native.Context(0, library=native.XCHECK_LIB_PATH) if not os.environ.get("RM_LIB") else None  # the wavefront pipeline: the tests' cross-check build
xh = xctx.create_scene(sc) if xctx else None
for name, flags in (("jump", F | NO), ("nojump", F | NO | NJ), ("wavefront", F | NO | abi.RM_RENDER_WAVEFRONT)):
    c, hc = (xctx, xh) if name == "wavefront" else (ctx, h)
    if c is None:
        continue
    fb = c.create_framebuffer(W, H)
    for nz in noises:
        c.render_sample(hc, fb, J.uniforms_from_schema(schema, nz), None, flags)
    planes[name] = [fb.download(k) for k in range(3)]
    fb.destroy()
for other in [o for o in ("nojump", "wavefront") if o in planes]:
    for k in range(3):
        a, b = planes["jump"][k], planes[other][k]
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        print(f"jump vs {other}: plane {k}: {int((~same).sum())} of {same.size} values differ")
fb = ctx.create_framebuffer(W, H)
u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
for name, flags in (("c3b jump", F | NO), ("c3b nojump", F | NO | NJ)):
    ctx.render_timed(h, fb, u, 3, None, flags)
    ms = sorted(ctx.render_timed(h, fb, u, 10, None, flags) for _ in range(3))
    print(f"{name}: {ms[0]:.4f} {ms[1]:.4f} {ms[2]:.4f} ms")
schema_p = J.make_schema(sc, W, H, counts=(256,), render_mode="preview", position=(0, 0, -2.5))
up = J.uniforms_from_schema(schema_p, (0.5, 1 / 3))
ctx.render_timed(h, fb, up, 3, None, F)
print("c3a preview: %.4f ms" % min(ctx.render_timed(h, fb, up, 10, None, F) for _ in range(3)))
fb.destroy(); h.destroy()
for wl, sc, kw in (("live", S.SphereGridFractal(), dict(width=1280, height=720, counts=(128, 128, 64, 32, 32), render_mode="full", position=(0, 0, 0))),
                   ("menger", S.MengerSponge(), dict(width=1920, height=1080, counts=(128, 64), render_mode="full", position=(0, 0, -3.0), lights=GC.LIGHT)),
                   ("kifs", S.KifsTree(), dict(width=1920, height=1080, counts=(128, 64), render_mode="full", position=(0, 0, -3.0), lights=GC.LIGHT)),
                   ("c4/8", S.csg64(), dict(width=4096, height=512, counts=(128,), render_mode="full", position=(0, 0, -5.0), lights=GC.LIGHT))):
    schema = J.make_schema(sc, **kw); h = ctx.create_scene(sc); fb = ctx.create_framebuffer(kw["width"], kw["height"])
    u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
    ctx.render_timed(h, fb, u, 1, None, F)
    print(f"{wl} {min(ctx.render_timed(h, fb, u, 3, None, F) for _ in range(3)):.3f} ms")
    fb.destroy(); h.destroy()
