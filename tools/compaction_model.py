"""Wave-steps of the C3b camera-ray march under different compaction scopes (model: cost = wave-steps,
every step costs the same).  Input: per-pixel settle step from the CAST_STEPS probe (fast build, default eps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S
ctx = native.Context(0)
sc = S.Mandelbulb(); h = ctx.create_scene(sc)
W, H = 3840, 2160
schema = J.make_schema(sc, W, H, counts=(256,), render_mode="full", position=(0, 0, -2.5))
u = J.uniforms_from_schema(schema, (0.5, 1/3))
cam = ctx.probe_camera(u, W, H)
rays = np.concatenate([cam[..., 0:3], cam[..., 4:7]], -1).reshape(-1, 6)
st = ctx.probe(h, abi.RM_PROBE_CAST_STEPS, rays, 256.0, 1).reshape(H, W).astype(np.int64)
st = np.minimum(st + 1, 256)  # steps executed
print("lane-steps/64 (ideal)", st.sum() / 64 / 1e6, "M wave-steps")
def tiles(a, n):
    Hh = (H // n) * n; Ww = (W // n) * n
    return a[:Hh, :Ww].reshape(Hh // n, n, Ww // n, n).transpose(0, 2, 1, 3).reshape(-1, n * n)
t8 = tiles(st, 8)
print("8x8 wave, lane-level settle", t8.max(-1).sum() / 1e6)
for n, K in ((16, 16), (32, 16), (64, 16)):
    t = tiles(st, n)
    cost = 0
    for k in range(0, 256, K):
        active = (t > k).sum(-1)
        cost += (np.ceil(active / 64) * np.minimum(K, np.maximum(t.max(-1) - k, 0))).sum()
    print(f"{n}x{n} workgroup compaction every {K}", cost / 1e6)
hist = np.bincount(st.ravel(), minlength=257)
print("fraction of lane-steps spent beyond step 64/128/192:", [float((np.maximum(st - k, 0)).sum() / st.sum()) for k in (64, 128, 192)])
print("fraction of rays running the full 256:", float((st >= 256).mean()))
