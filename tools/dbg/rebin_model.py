"""Would culling pay in the bounces after the first if a workgroup's rays were re-ordered by culling cell at every repack?  (round 5; VERDICT r4's
"cut at bounces with rays re-binned by cell", at workgroup granularity: the compaction already moves rays between lanes through LDS.)

A numpy model of one 16 x 32-pixel tile of the C5 job (CSG-64, camera at z = -5): the camera rays are marched, the hit points bounce diffusely, and the
second-bounce rays are marched 64 steps.  At every step the rays still moving are packed into waves of 64 (a) in pixel order, as the kernel's
compaction does, (b) sorted by the Morton code of their level-0 cell; the row list of a wave is the union of its rays' cells' lists, from the
library's own rule (native.cull_cell = rm_debug_cull_cell, host arithmetic).  Prints rows folded per ray and step: every row (64), the ray's own
list (the ideal), (a), (b).   python tools/dbg/rebin_model.py [tiles...]"""
import importlib
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S = importlib.import_module("raymarching-engine_amd.scene")
N = importlib.import_module("raymarching-engine_amd.native")
f32 = np.float32
sc = S.csg64()
C = np.array([n.center if not isinstance(n, tuple) else n[3] for n in sc._nodes], np.float64)
R = np.array([(n.size if not isinstance(n, tuple) else n[4])[0] for n in sc._nodes], np.float64)
K = 0.2
HALF, NCELL = 2.2, 128
REPACK = int(os.environ.get("REPACK", "8"))
CELL = 2 * HALF / NCELL
_lists = {}


def fold(p):
    q = p[:, None, :] - C[None].astype(f32)
    di = np.sqrt((q * q).sum(-1, dtype=f32)).astype(f32) - R[None].astype(f32)
    d = di[:, 0].copy()
    for i in range(1, len(R)):
        t = di[:, i] - d
        h = np.clip(f32(0.5) + f32(0.5 / K) * t, f32(0), f32(1)).astype(f32)
        d = (di[:, i] - h * (t + f32(K) * (f32(1) - h))).astype(f32)
    return d


def cell_of(p):
    c = np.floor((p + HALF) / CELL).astype(np.int64)
    inside = ((c >= 0) & (c < NCELL)).all(-1)
    return c, inside


def list_of(c):
    key = tuple(int(v) for v in c)
    if key not in _lists:
        centre = [-HALF + (v + 0.5) * CELL for v in key]
        _lists[key] = np.array(N.cull_cell(sc, centre, 0.8660254 * CELL * 1.002, 1e-6), bool)
    return _lists[key]


def morton(c):
    m = np.zeros(len(c), np.int64)
    for b in range(7):
        for a in range(3):
            m |= ((c[:, a] >> b) & 1) << (3 * b + a)
    return m


def tile(tx, ty, W=8192, fov=1.5, seed=0):
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[ty:ty + 32, tx:tx + 16]
    th = math.tan(fov / 2)
    d = np.stack([((xs + 0.5) / W * 2 - 1) * th, ((ys + 0.5) / W * 2 - 1) * th, np.ones_like(xs, float)], -1).reshape(-1, 3)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f32)
    p = np.tile(np.array([0, 0, -5], f32), (len(d), 1))
    for _ in range(128):
        p = (p + d * fold(p)[:, None]).astype(f32)
    hit = fold(p) < 1e-3
    if hit.sum() < 64:
        return None
    p = p[hit]
    e = f32(1e-4)
    n = np.stack([fold(p + np.array(a, f32) * e) - fold(p) for a in ((1, 0, 0), (0, 1, 0), (0, 0, 1))], -1)
    n /= np.linalg.norm(n, axis=1, keepdims=True) + 1e-30
    v = rng.normal(size=n.shape)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v = n + v  # cosine-weighted
    v = (v / (np.linalg.norm(v, axis=1, keepdims=True) + 1e-30)).astype(f32)
    q = (p + n.astype(f32) * f32(1e-3)).astype(f32)
    alive = np.ones(len(q), bool)
    tot = np.zeros(5)
    steps = 0
    group = None  # wave of every ray, fixed at the last repack
    for s in range(64):
        dist = fold(q)
        moving = alive & (np.abs(q).max(-1) < 8.0) & (dist > 1e-6)
        idx = np.flatnonzero(moving)
        if idx.size == 0:
            break
        c, inside = cell_of(q[idx].astype(np.float64))
        lists = np.array([list_of(ci) if ok else np.ones(64, bool) for ci, ok in zip(c, inside)])
        own = lists.sum(1).mean()

        def waves(order):
            u = [lists[order[i:i + 64]].any(0).sum() * len(order[i:i + 64]) for i in range(0, len(order), 64)]
            return sum(u) / len(order)
        a = waves(np.arange(len(idx)))
        key = np.where(inside, morton(np.clip(c, 0, NCELL - 1)), (1 << 40))
        order = np.argsort(key, kind="stable")
        b = waves(order)
        if s % REPACK == 0:  # the kernel's cadence: the waves formed now stay until the next repack (rays that stop leave their lanes idle)
            group = np.full(len(q), -1)
            group[idx[order]] = np.arange(len(idx)) // 64
        g = group[idx]
        e = sum(lists[g == w].any(0).sum() * (g == w).sum() for w in np.unique(g)) / len(idx)
        tot += np.array([64.0, own, a, b, e]) * len(idx)
        steps += len(idx)
        q = (q + v * dist[:, None]).astype(f32)
        alive = moving
    return tot / steps, steps, int(hit.sum())


if __name__ == "__main__":
    tiles = [(4096, 4096), (3600, 4300), (4600, 3700), (3200, 3300), (5000, 5000), (4096, 3000)]
    print(f"tile            rays  ray-steps   every row   own list   waves in pixel order   waves sorted by cell   sorted every {REPACK} steps only")
    acc, n = np.zeros(5), 0
    for t in tiles:
        r = tile(*t)
        if r is None:
            print(t, "misses the scene")
            continue
        m, steps, rays = r
        print(f"{str(t):14s} {rays:5d} {steps:9d} {m[0]:10.1f} {m[1]:10.1f} {m[2]:18.1f} {m[3]:22.1f} {m[4]:22.1f}")
        acc += m * steps
        n += steps
    print("all", (acc / n).round(1))
