#!/usr/bin/env python3
"""The small kernels around the pixel kernel, for `rocprofv3 --kernel-trace --stats -- python3 tools/small_kernels_probe.py`:
the culling grid's build (CSG-64, one per new scene) and the tile sort (rm_order_*; C5 has 131 072 tiles).
Also prints the wall time of a C4 job that gets a NEW scene every frame against the same scene every frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
F = abi.RM_RENDER_FAST
def jittered(seed):  # CSG-64 with other centres: a new scene of the same shape
    rng = np.random.default_rng(seed); t = S.CsgScene()
    for i in range(64):
        t.smooth_union(0.2)
        t.sphere(((i % 4 - 1.5) * 0.9 + rng.uniform(-0.2, 0.2), ((i // 4) % 4 - 1.5) * 0.9 + rng.uniform(-0.2, 0.2), (i // 16 - 1.5) * 0.9 + rng.uniform(-0.2, 0.2)), float(rng.uniform(0.25, 0.45)))
    return t
kw = dict(width=4096, height=4096, counts=(128,), render_mode="full", position=(0, 0, -5.0), lights=GC.LIGHT)
fb = ctx.create_framebuffer(4096, 4096)
def run(new_scene_every_frame, frames=8):
    scenes = [jittered(100 + (i if new_scene_every_frame else 0)) for i in range(frames + 2)]
    schema = J.make_schema(scenes[0], **kw); u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
    hs = []
    for i in range(2):  # warm-up
        h = ctx.create_scene(scenes[i]); ctx.render_sample(h, fb, u, None, F); hs.append(h)
    ctx.sync(); t0 = time.perf_counter()
    for i in range(2, frames + 2):
        h = ctx.create_scene(scenes[i]) if new_scene_every_frame else hs[0]
        ctx.render_sample(h, fb, u, None, F)
        if new_scene_every_frame: hs.append(h)
    ctx.sync(); dt = (time.perf_counter() - t0) / frames * 1e3
    for h in hs: h.destroy()
    return dt
for mp in (0, None):
    if mp is not None: ctx.set_cull_min_pixels(mp)
    else: ctx.set_cull_min_pixels(4 << 20)
    print(f"cull_min_pixels {'default (4 Mi)' if mp is None else mp}: C4 same scene every frame {run(False):.2f} ms/frame, NEW scene every frame {run(True):.2f} ms/frame, stats {ctx.cull_stats()}")
os.environ["RM_NO_CULL"] = "1"
print(f"no grid at all (RM_NO_CULL=1): new scene every frame {run(True):.2f} ms/frame")
os.environ["RM_NO_CULL"] = "0"
fb.destroy()
# C5: 131 072 tiles through the sort
sc = S.csg64(); h = ctx.create_scene(sc)
schema = J.make_schema(sc, width=8192, height=8192, counts=(128, 64, 64), render_mode="full", position=(0, 0, -5.0), lights=[J.point_light((2.0, 3.0, -4.0), size=0.3)])
fb = ctx.create_framebuffer(8192, 8192); u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
print("C5 ms/sample:", ctx.render_timed(h, fb, u, 3, None, F | abi.RM_RENDER_NO_OVERLAP))
