#!/usr/bin/env python3
"""Table scenes (CSG-64) by library variant: python tools/table_variants.py default tools/_exp_x.so ...
ms per sample of C4 (4096^2 full frame), one 8-way shard of it, one 8-way shard of C5, C5, and C2 (the pixel kernel: the wavefront pipeline
is in the tests' cross-check build only since round 5 -- tools/time_all.py times it)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) >= 2 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        env = dict(os.environ, RM_LIB=os.path.abspath(lib) if lib not in ("default", "nocull") else "", RM_NO_CULL="1" if lib == "nocull" else "0")  # nocull: no culling grid for the tables (rm_api.hip table_cull_grid)
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print("==", lib); print(r.stdout.strip()); print(r.stderr.strip()[-600:])
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S, shard
ctx = native.Context(0)
F, MK, WF = abi.RM_RENDER_FAST, abi.RM_RENDER_MEGAKERNEL, abi.RM_RENDER_WAVEFRONT
soft = [J.point_light((2.0, 3.0, -4.0), size=0.3)]
cases = [("c4 full", S.csg64(), dict(width=4096, height=4096, counts=(128,), render_mode="full", position=(0, 0, -5.0), lights=GC.LIGHT), None, (MK,)),
         ("c4 shard/8", S.csg64(), dict(width=4096, height=4096, counts=(128,), render_mode="full", position=(0, 0, -5.0), lights=GC.LIGHT), 8, (MK,)),
         ("c5 shard/8", S.csg64(), dict(width=8192, height=8192, counts=(128, 64, 64), render_mode="full", position=(0, 0, -5.0), lights=soft), 8, (MK,)),
         ("c5 full", S.csg64(), dict(width=8192, height=8192, counts=(128, 64, 64), render_mode="full", position=(0, 0, -5.0), lights=soft), None, (MK,)),
         ("c2", S.single_sphere(), dict(width=1920, height=1080, counts=(128,), render_mode="preview", position=(0, 0, -3.0)), None, (MK,)),
         ("csg_blocks (192 rows, hard operators) 1080p", S.csg_blocks(), dict(width=1920, height=1080, counts=(128, 64), render_mode="full", position=(0.3, 0.2, -6.0), lights=GC.LIGHT), None, (MK,)),
         ("csg_blocks preview 1080p", S.csg_blocks(), dict(width=1920, height=1080, counts=(128,), render_mode="preview", position=(0.3, 0.2, -6.0)), None, (MK,)),
         ("csg_mixed 1080p", GC.build_scene("csg_mixed"), dict(width=1920, height=1080, counts=(128, 64), render_mode="full", position=(0.3, 0.2, -4.0), lights=GC.LIGHT), None, (MK,))]
def smooth_table(rows, boxes, seed=11):  # mostly smooth unions of several radii (round 4: the general fold culls their far rows too)
    import numpy as np
    rng = np.random.default_rng(seed); t = S.CsgScene()
    for i in range(rows):
        t.smooth_union(float(np.float32(rng.uniform(0.05, 0.4)))) if rng.uniform() < 0.9 or i == 0 else t.union()
        c = rng.uniform(-2, 2, 3)
        t.box(c, rng.uniform(0.1, 0.5, 3)) if boxes and rng.uniform() < 0.4 else t.sphere(c, float(rng.uniform(0.2, 0.6)))
    return t
cases += [("smooth spheres, 64 rows of several radii, 1080p", smooth_table(64, False), dict(width=1920, height=1080, counts=(128, 64), render_mode="full", position=(0.3, 0.2, -6.0), lights=GC.LIGHT), None, (MK,)),
          ("smooth spheres and boxes, 96 rows, 1080p", smooth_table(96, True), dict(width=1920, height=1080, counts=(128, 64), render_mode="full", position=(0.3, 0.2, -6.0), lights=GC.LIGHT), None, (MK,))]
for name, sc, kw, parts, impls in cases:
    schema = J.make_schema(sc, **kw); h = ctx.create_scene(sc)
    fb = ctx.create_framebuffer(kw["width"], kw["height"]) if parts is None else ctx.create_striped_framebuffer(kw["width"], kw["height"], shard.STRIPE_ROWS, parts, 0)
    u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
    for impl in impls:
        ctx.render_timed(h, fb, u, 2, None, F | impl | abi.RM_RENDER_NO_OVERLAP)
        ms = min(ctx.render_timed(h, fb, u, 3, None, F | impl | abi.RM_RENDER_NO_OVERLAP) for _ in range(2))
        print(f"{name} {'wavefront' if impl == WF else 'pixel kernel'}: {ms:.3f} ms")
    fb.destroy(); h.destroy()
