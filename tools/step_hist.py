"""How many castRay steps until a ray's position stops changing (C3b camera rays, 4K frame)."""
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
cam = ctx.probe_camera(u, W, H)              # jitter-free rays
rays = np.concatenate([cam[..., 0:3], cam[..., 4:7]], -1).reshape(-1, 6)
eps = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
ctx.set_retire_eps(eps); print("eps", eps)
st = ctx.probe(h, abi.RM_PROBE_CAST_STEPS, rays, 256.0, 1).reshape(H, W)
end = ctx.probe(h, abi.RM_PROBE_CAST_RAY, rays, 256.0, 1).reshape(H, W, 3)
hit = np.isfinite(end).all(-1)
print("hit fraction", hit.mean())
for name, m in (("sky", ~hit), ("hit", hit)):
    v = st[m]
    print(name, "n", v.size, "mean", v.mean(), "pcts 50/90/99", np.percentile(v, [50, 90, 99]), "frac never settled", np.mean(v >= 256))
# per-wave (8x8 tile) max vs mean over hit tiles
t = st.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(H // 8, W // 8, 64)
th = hit.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(H // 8, W // 8, 64)
anyhit = th.any(-1)
print("tiles with a hit", anyhit.mean(), "mean of tile-max steps", t.max(-1)[anyhit].mean(), "mean of lane steps in those tiles", t[anyhit].mean())
print("tile-max histogram", np.histogram(t.max(-1)[anyhit], bins=[0, 32, 64, 96, 128, 192, 255, 257])[0])
print("lane histogram (hit lanes)", np.histogram(st[hit], bins=[0, 32, 64, 96, 128, 192, 255, 257])[0])
