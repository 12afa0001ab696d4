"""Model of row culling for smooth-union sphere tables (round 3, before any kernel was written).

A row whose distance is at least k above the running value of the fold is an exact no-op (h clamps), so a grid over the
scene can hold, per cell, the rows that can matter anywhere in the cell: row i is dropped when
    d_i(centre) - R  >=  min_{j<i} (d_j(centre) + R) + k + margin        (R = half the cell's diagonal; the shapes are 1-Lipschitz).
This marches the camera rays and the shadow rays of sample tiles of BASELINE's C4 frame in float32 numpy and counts, per wave
(8x8 pixels) and step: rows in the union of the lanes' cell masks, rows of the fullest lane, and the ideal (rows that matter at
the lane's own point).  python tools/dbg/cull_model.py [grid ...]
"""
import importlib
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
S = importlib.import_module("raymarching-engine_amd.scene")

f32 = np.float32
OUTER = float(os.environ.get("OUTER", "0"))
OUTER_N = int(os.environ.get("OUTER_N", "32"))


def spheres_of(sc):
    c, r = [], []
    for n in sc._nodes:
        c.append(n[3] if isinstance(n, tuple) else n.center)
        r.append((n[4] if isinstance(n, tuple) else n.size)[0])
    return np.array(c, dtype=np.float64), np.array(r, dtype=np.float64)


def fold(p, C, R, k):
    """the smooth-union fold at points p [n,3] (float32); returns d and the per-row raw distances"""
    q = p[:, None, :] - C[None, :, :].astype(f32)
    di = np.sqrt((q * q).sum(-1, dtype=f32)).astype(f32) - R[None, :].astype(f32)
    d = di[:, 0].copy()
    for i in range(1, C.shape[0]):
        t = di[:, i] - d
        g = np.clip(f32(0.5) - f32(0.5 / k) * t, f32(0), f32(1)).astype(f32)
        d = (d + g * (t - f32(k) + f32(k) * g)).astype(f32)
    return d, di


def build_grid(C, R, k, n, lo, hi):
    cell = (hi - lo) / n
    rad = 0.5 * math.sqrt(3.0) * cell
    ax = lo + (np.arange(n) + 0.5) * cell
    gx, gy, gz = np.meshgrid(ax, ax, ax, indexing="ij")
    cen = np.stack([gx, gy, gz], -1).reshape(-1, 3)
    d = np.sqrt(((cen[:, None, :] - C[None]) ** 2).sum(-1)) - R[None]
    keep = np.ones(d.shape, dtype=bool)
    upper = d[:, 0] + rad
    margin = 1e-4 * (1.0 + np.abs(cen).max())
    for i in range(1, C.shape[0]):
        keep[:, i] = ~(d[:, i] - rad >= upper + k + margin)
        upper = np.minimum(upper, d[:, i] + rad)
    return keep.reshape(n, n, n, -1), cell


def main():
    grids = [int(a) for a in sys.argv[1:]] or [16, 32, 64]
    sc = S.csg64()
    C, R = spheres_of(sc)
    k = 0.2
    lo = float((C - R[:, None]).min()) - 0.3
    hi = float((C + R[:, None]).max()) + 0.3
    W = H = 4096
    fov, cam = 1.5, np.array([0.0, 0.0, -5.0], dtype=f32)
    light = np.array([2.0, 3.0, -4.0], dtype=f32)
    rng = np.random.default_rng(5)
    th = math.tan(fov / 2)
    tiles = [(int(rng.integers(0, W // 16)), int(rng.integers(0, H // 32))) for _ in range(160)]
    stats = {n: dict(union=0, maxlane=0, ideal=0, waves=0, outside=0, meanlane=0.0) for n in grids}
    masks = {n: build_grid(C, R, k, n, lo, hi) for n in grids}
    outer = build_grid(C, R, k, OUTER_N, -OUTER, OUTER) if OUTER else None
    if outer:
        print(f"outer grid {OUTER_N}: +-{OUTER}, mean rows kept per cell {outer[0].sum(-1).mean():.1f}, cell {outer[1]:.3f}")
    for n in grids:
        print(f"grid {n}: mean rows kept per cell {masks[n][0].sum(-1).mean():.1f}, cell {masks[n][1]:.3f}")
    total_steps = 0
    for (tx, ty) in tiles:
        xs, ys = np.meshgrid(np.arange(16) + tx * 16, np.arange(32) + ty * 32, indexing="xy")
        px = ((xs + 0.5) / W * 2 - 1) * th
        py = ((ys + 0.5) / H * 2 - 1) * th
        d = np.stack([px, py, np.ones_like(px)], -1).reshape(-1, 3)
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f32)
        p = np.tile(cam, (d.shape[0], 1)).astype(f32)
        wave = ((ys // 8 - ty * 4) * 2 + (xs // 8 - tx * 2)).reshape(-1)
        for phase in range(2):
            live = np.ones(p.shape[0], dtype=bool)
            for step in range(128):
                dist, di = fold(p, C, R, k)
                # escaping rays: jumped in the kernel
                r2 = (p * p).sum(-1)
                esc = (r2 > 25.0) & ((p * d).sum(-1) >= 0)
                live &= ~esc
                if not live.any():
                    break
                # ideal: rows that matter at the lane's own point (running value of the fold is above d_i - k)
                run = di[:, 0].copy()
                matter = np.ones(di.shape, dtype=bool)
                for i in range(1, C.shape[0]):
                    matter[:, i] = di[:, i] - run < k
                    t = di[:, i] - run
                    g = np.clip(0.5 - 0.5 / k * t, 0, 1)
                    run = run + g * (t - k + k * g)
                for n in grids:
                    keep, cell = masks[n]
                    idx = np.floor((p - f32(lo)) / f32(cell)).astype(int)
                    inside = ((idx >= 0) & (idx < n)).all(-1)
                    idx = np.clip(idx, 0, n - 1)
                    m = keep[idx[:, 0], idx[:, 1], idx[:, 2]].copy()
                    if OUTER:  # a second, coarse grid around the first
                        keep2, cell2 = outer
                        idx2 = np.floor((p - f32(-OUTER)) / f32(cell2)).astype(int)
                        inside2 = ((idx2 >= 0) & (idx2 < keep2.shape[0])).all(-1)
                        idx2 = np.clip(idx2, 0, keep2.shape[0] - 1)
                        m2 = keep2[idx2[:, 0], idx2[:, 1], idx2[:, 2]].copy()
                        m2[~inside2] = True
                        m[~inside] = m2[~inside]
                        inside = inside | inside2
                    else:
                        m[~inside] = True
                    st = stats[n]
                    for w in range(8):
                        sel = (wave == w) & live
                        if not sel.any():
                            continue
                        st["waves"] += 1
                        st["union"] += int(m[sel].any(0).sum())
                        st["maxlane"] += int(m[sel].sum(-1).max())
                        st["meanlane"] += float(m[sel].sum(-1).mean())
                        st["ideal"] += float(matter[sel].sum(-1).mean())
                        st["outside"] += int((~inside[sel]).any())
                        # the kernel's union: the first lane's set, then the set of the first lane with a row outside it, CAP times; then every row
                        ms = m[sel]
                        u = ms[0].copy()
                        it = 0
                        while True:
                            more = (ms & ~u).any(-1)
                            if not more.any():
                                break
                            it += 1
                            u |= ms[np.argmax(more)]
                        st.setdefault("iters", []).append(it)
                q = (p + d * dist[:, None]).astype(f32)
                same = (q.view(np.uint32) == p.view(np.uint32)).all(-1)
                p = np.where(live[:, None], q, p)
                live &= ~same
                total_steps += 1
                if not live.any():
                    break
            # shadow rays from the end points
            hit = np.isfinite(p).all(-1) & ((p * p).sum(-1) < 25.0)
            if not hit.any():
                break
            p = p[hit]
            wave = wave[hit]
            to = light[None] - p
            d = (to / np.linalg.norm(to, axis=1, keepdims=True)).astype(f32)
            p = (p + d * f32(0.001)).astype(f32)
    for n in grids:
        st = stats[n]
        w = max(1, st["waves"])
        its = np.array(st.get("iters", [0]))
        print(f"grid {n:3d}: union rounds needed per wave-step: mean {its.mean():.2f}, share needing more than 1 / 2 / 3 / 5 / 8: "
              + " / ".join(f"{(its > c).mean():.3f}" for c in (1, 2, 3, 5, 8)))
        print(f"grid {n:3d}: wave-steps {w}; rows per wave-step: union of the lanes' masks {st['union'] / w:.1f}, fullest lane {st['maxlane'] / w:.1f}, "
              f"mean lane {st['meanlane'] / w:.1f}, ideal per lane {st['ideal'] / w:.1f}; wave-steps with a lane outside the grid {st['outside'] / w:.3f}")


if __name__ == "__main__":
    main()
