"""Second model of row culling on BASELINE's C4 frame (round 3): where the evaluations are -- camera march, shadow rays of
pixels that hit the scene, shadow rays of sky pixels (cast from 1e6 away in a random direction, raymarcher.frag:279,354-362) --
and what the wave's union of cell masks holds in each class.  8x8-pixel waves, no compaction.  python tools/dbg/cull_model2.py"""
import importlib, math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cull_model as M
f32 = np.float32
N1, N2, OUT = int(os.environ.get("N1", 32)), int(os.environ.get("N2", 24)), float(os.environ.get("OUTER", 3.0))
CAP = int(os.environ.get("CAP", 3))
BINS = [(0, 2.3), (2.3, 3), (3, 4), (4, 5.5), (5.5, 7), (7, 10), (10, 14), (14, 1e9)]


def main():
    sc = M.S.csg64()
    C, R = M.spheres_of(sc)
    k = 0.2
    widest = float(((C + R[:, None]).max(0) - (C - R[:, None]).min(0)).max())
    pad = k + 0.05 * widest + 1e-3
    lo = float((C - R[:, None]).min()) - pad
    hi = float((C + R[:, None]).max()) + pad
    half = 0.5 * OUT * (hi - lo)
    inner, cell1 = M.build_grid(C, R, k, N1, lo, hi)
    outer, cell2 = M.build_grid(C, R, k, N2, -half, half)
    far_r2 = (2 * (np.sqrt((C * C).sum(-1)).max() + R.max() + 0.25 * k * 63) + 1) ** 2
    print(f"inner {N1}^3 over [{lo:.2f}, {hi:.2f}] (cell {cell1:.3f}, {inner.sum(-1).mean():.1f} rows per cell), outer {N2}^3 over +-{half:.2f} (cell {cell2:.3f}, {outer.sum(-1).mean():.1f} rows per cell), far radius {math.sqrt(far_r2):.1f}")
    W = H = 4096
    th = math.tan(0.75)
    cam = np.array([0, 0, -5.0], f32)
    light = np.array([2.0, 3.0, -4.0], f32)
    rng = np.random.default_rng(7)
    tiles = [(int(rng.integers(0, W // 8)), int(rng.integers(0, H // 8))) for _ in range(int(os.environ.get("TILES", 700)))]
    stat = {}

    def masks(p):
        i1 = np.floor((p - f32(lo)) / f32(cell1)).astype(int)
        in1 = ((i1 >= 0) & (i1 < N1)).all(-1)
        i1 = np.clip(i1, 0, N1 - 1)
        m = inner[i1[:, 0], i1[:, 1], i1[:, 2]].copy()
        i2 = np.floor((p + f32(half)) / f32(cell2)).astype(int)
        in2 = ((i2 >= 0) & (i2 < N2)).all(-1)
        i2 = np.clip(i2, 0, N2 - 1)
        m2 = outer[i2[:, 0], i2[:, 1], i2[:, 2]].copy()
        m2[~in2] = True
        m[~in1] = m2[~in1]
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
                u = ms[0].copy()
                it = 0
                while (ms & ~u).any():
                    it += 1
                    if it > CAP:
                        u[:] = True
                        break
                    u |= ms[np.argmax((ms & ~u).any(-1))]
                st = stat.setdefault(int(c), dict(wave_steps=0, lane_steps=0, rows=0, fallback=0, lane_rows=0, max_lane=0))
                st["wave_steps"] += 1
                st["lane_steps"] += int(sel.sum())
                st["rows"] += int(u.sum())
                st["fallback"] += int(it > CAP)
                st["lane_rows"] += int(ms.sum())
                st["max_lane"] += int(ms.sum(-1).max())
                rr = np.sqrt((p[sel].astype(np.float64) ** 2).sum(-1))
                hb = st.setdefault("hist", np.zeros((8, 2)))
                for b, (a0, a1) in enumerate(BINS):
                    q = (rr >= a0) & (rr < a1)
                    hb[b, 0] += q.sum()
                    hb[b, 1] += ms[q].sum()
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
        start = (start + sd * f32(0.001)).astype(f32) if False else start
        march(start.astype(f32), sd, 128, np.where(sky, 2, 1))
    names = {0: "camera march", 1: "shadow rays of pixels that hit", 2: "shadow rays of sky pixels"}
    tot = sum(s["wave_steps"] for s in stat.values())
    for c, s in sorted(stat.items()):
        w = s["wave_steps"]
        print(f"{names[c]:34s}: {w / len(tiles):6.1f} wave-steps per wave ({100 * w / tot:4.1f} %), lanes active {s['lane_steps'] / w:5.1f}; rows per wave-step: union (cap {CAP}) {s['rows'] / w:5.1f}, "
              f"fullest lane {s['max_lane'] / w:5.1f}, mean lane {s['lane_rows'] / s['lane_steps']:5.1f}; fallback to every row {100 * s['fallback'] / w:4.1f} %")
        print("      lane-steps by |p| " + ", ".join(f"[{a0:g},{a1:g}): {100 * h[0] / s['lane_steps']:.0f} % x {h[1] / max(1, h[0]):.0f} rows" for (a0, a1), h in zip(BINS, s["hist"])))


main()
