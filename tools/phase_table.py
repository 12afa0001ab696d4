#!/usr/bin/env python3
"""Round 3: cumulative cost of the pixel kernel's phases on a table job, with and without row culling (the smooth-union experiment ran on C4 / C5; csg_blocks is a table the culling applies to).
python tools/phase_cost.py && python tools/phase_table.py [c4|c5s]   (the RM_DIAG_STOP builds of tools/phase_cost.py)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" not in sys.argv:
    wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
    for n in (1, 2, 3, 4, 5, 0):
        env = dict(os.environ, RM_LIB=os.path.join(ROOT, "tools", f"_exp_stop{n}.so") if n else "")
        r = subprocess.run([sys.executable, __file__, "--child", wl], env=env, capture_output=True, text=True)
        print(f"stop {n}:", r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else "")
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S, shard
ctx = native.Context(0)
F, MK, NC = abi.RM_RENDER_FAST, abi.RM_RENDER_MEGAKERNEL, abi.RM_RENDER_NO_CULL
wl = sys.argv[-1]
soft = [J.point_light((2.0, 3.0, -4.0), size=0.3)]
kw = dict(width=4096, height=4096, counts=(128,), render_mode="full", position=(0, 0, -5.0), lights=GC.LIGHT) if wl == "c4" else \
     dict(width=8192, height=8192, counts=(128, 64, 64), render_mode="full", position=(0, 0, -5.0), lights=soft)
sc = S.csg_blocks() if wl == "blocks" else S.csg64()
if wl == "blocks":
    kw = dict(width=1920, height=1080, counts=(128, 64), render_mode="full", position=(0.3, 0.2, -6.0), lights=GC.LIGHT)
schema = J.make_schema(sc, **kw); h = ctx.create_scene(sc)
fb = ctx.create_framebuffer(kw["width"], kw["height"]) if wl in ("c4", "blocks") else ctx.create_striped_framebuffer(kw["width"], kw["height"], shard.STRIPE_ROWS, 8, 0)
u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
out = []
for flags in (F | MK, F | MK | NC):
    ctx.render_timed(h, fb, u, 2, None, flags | abi.RM_RENDER_NO_OVERLAP)
    out.append(min(ctx.render_timed(h, fb, u, 3, None, flags | abi.RM_RENDER_NO_OVERLAP) for _ in range(2)))
print(f"culled {out[0]:.3f} ms, every row {out[1]:.3f} ms")
