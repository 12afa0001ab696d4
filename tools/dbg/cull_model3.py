"""Third model of row culling (round 3): nested grids ("clipmap": level l covers the inner box scaled by 2^l, the same cell count)
and the pairwise bound for sphere rows -- d_i - d_j has the Lipschitz constant |c_i - c_j| / sqrt(|p - c_i| |p - c_j|) <= 2, so far
cells may be large.  Per-lane row counts by class of ray on BASELINE's C4 frame.  python tools/dbg/cull_model3.py"""
import importlib, math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cull_model as M
f32 = np.float32
N, LEVELS = int(os.environ.get("N", 32)), int(os.environ.get("LEVELS", 5))
PAIR = int(os.environ.get("PAIR", 1))


def build(C, R, k, n, lo, hi):
    cell = (hi - lo) / n
    rad = 0.5 * math.sqrt(3.0) * cell * 1.001
    ax = lo + (np.arange(n) + 0.5) * cell
    gx, gy, gz = np.meshgrid(ax, ax, ax, indexing="ij")
    cen = np.stack([gx, gy, gz], -1).reshape(-1, 3)
    a = np.sqrt(((cen[:, None, :] - C[None]) ** 2).sum(-1))  # distance to the centres
    d = a - R[None]
    keep = np.ones(d.shape, dtype=bool)
    upper = d[:, 0] + rad
    best = np.zeros(len(cen), int)  # the nearest previous row at the cell's centre
    margin = 1e-3 * (1.0 + np.abs(cen).max())
    ar = np.arange(len(cen))
    for i in range(1, C.shape[0]):
        cull = d[:, i] - rad >= upper + k + margin
        if PAIR:
            s = np.sqrt(((C[i][None] - C[best]) ** 2).sum(-1))
            ai, aj = a[:, i] - rad, a[ar, best] - rad
            lip = np.where((ai > 0) & (aj > 0), np.minimum(2.0, s / np.sqrt(np.maximum(ai * aj, 1e-30))), 2.0)
            cull |= d[:, i] - d[ar, best] - rad * lip >= k + margin
        keep[:, i] = ~cull
        upper = np.minimum(upper, d[:, i] + rad)
        best = np.where(d[:, i] < d[ar, best], i, best)
    return keep.reshape(n, n, n, -1), cell


def main():
    sc = M.S.csg64()
    C, R = M.spheres_of(sc)
    k = 0.2
    widest = float(((C + R[:, None]).max(0) - (C - R[:, None]).min(0)).max())
    pad = k + 0.05 * widest + 1e-3
    half0 = max(abs(float((C - R[:, None]).min()) - pad), abs(float((C + R[:, None]).max()) + pad))
    levels = [build(C, R, k, N, -half0 * 2 ** l, half0 * 2 ** l) for l in range(LEVELS)]
    print(f"{LEVELS} levels of {N}^3, half-width {half0:.2f} x 2^l; rows per cell by level: " + ", ".join(f"{g.sum(-1).mean():.1f}" for g, _ in levels))
    far_r2 = (2 * (np.sqrt((C * C).sum(-1)).max() + R.max() + 0.25 * k * 63) + 1) ** 2
    W = H = 4096
    th = math.tan(0.75)
    cam = np.array([0, 0, -5.0], f32)
    light = np.array([2.0, 3.0, -4.0], f32)
    rng = np.random.default_rng(7)
    tiles = [(int(rng.integers(0, W // 8)), int(rng.integers(0, H // 8))) for _ in range(int(os.environ.get("TILES", 250)))]
    stat = {}

    def masks(p):
        m = np.ones((len(p), C.shape[0]), bool)
        mx = np.abs(p).max(-1)
        lvl = np.clip(np.ceil(np.log2(np.maximum(mx / half0, 1e-9))), 0, None).astype(int)
        for l in range(LEVELS):
            sel = lvl == l
            if not sel.any():
                continue
            g, cell = levels[l]
            idx = np.clip(np.floor((p[sel] + half0 * 2 ** l) / cell).astype(int), 0, N - 1)
            m[sel] = g[idx[:, 0], idx[:, 1], idx[:, 2]]
        return m

    def march(p, d, steps, cls):
        live = np.ones(len(p), bool)
        for _ in range(steps):
            r2 = (p.astype(np.float64) ** 2).sum(-1)
            esc = (r2 > far_r2) & ((p * d).sum(-1) >= 0) | ~np.isfinite(r2)
            live &= ~esc
            for c in np.unique(cls[live]):
                sel = live & (cls == c)
                ms = masks(p[sel])
                st = stat.setdefault(int(c), dict(wave_steps=0, lane_steps=0, union=0, lane_rows=0, max_lane=0))
                st["wave_steps"] += 1
                st["lane_steps"] += int(sel.sum())
                st["union"] += int(ms.any(0).sum())
                st["lane_rows"] += int(ms.sum())
                st["max_lane"] += int(ms.sum(-1).max())
            if not live.any():
                break
            dist, _ = M.fold(p, C, R, k)
            q = (p + d * dist[:, None]).astype(f32)
            same = (q.view(np.uint32) == p.view(np.uint32)).all(-1)
            p = np.where(live[:, None], q, p)
            live &= ~same
        return p

    for tx, ty in tiles:
        xs, ys = np.meshgrid(np.arange(8) + tx * 8, np.arange(8) + ty * 8, indexing="xy")
        d = np.stack([((xs + 0.5) / W * 2 - 1) * th, ((ys + 0.5) / H * 2 - 1) * th, np.ones(xs.shape)], -1).reshape(-1, 3)
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f32)
        p0 = np.tile(cam, (64, 1))
        p = march(p0.copy(), d, 128, np.zeros(64, int))
        sky = ~np.isfinite(p).all(-1) | ((p.astype(np.float64) ** 2).sum(-1) > far_r2)
        nd = rng.normal(0, 1, (64, 3))
        nd = (nd / np.linalg.norm(nd, axis=1, keepdims=True)).astype(f32)
        start = np.where(sky[:, None], p0 + nd * f32(1e6), p)
        to = light[None] - start
        sd = (to / np.linalg.norm(to, axis=1, keepdims=True)).astype(f32)
        march(start.astype(f32), sd, 128, np.where(sky, 2, 1))
    names = {0: "camera march", 1: "shadow rays of pixels that hit", 2: "shadow rays of sky pixels"}
    for c, s in sorted(stat.items()):
        w = s["wave_steps"]
        print(f"{names[c]:34s}: lane-steps per wave {s['lane_steps'] / len(tiles):6.0f}; rows: union of the wave {s['union'] / w:5.1f}, fullest lane {s['max_lane'] / w:5.1f}, mean lane {s['lane_rows'] / s['lane_steps']:5.1f}")


main()
