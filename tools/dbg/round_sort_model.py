#!/usr/bin/env python3
"""CPU model: how many Mandelbulb rounds a workgroup ISSUES (per wave and step: the deepest lane's count) against the rounds
its rays USE, for the camera march of headline-frame tiles, under (a) the kernel's repack (every 16 steps, stable compaction
when a wave can be freed), (b) a repack that also sorts the rays by the round count of their last evaluation, (c) a sort at
every step (bound).  float32 numpy restatement of the trig-free power-8 estimator (rounding differs from the kernel's)."""
import sys
import numpy as np
f = np.float32
def de(p):
    z = p.copy(); dr = np.ones(len(p), f); r2 = np.zeros(len(p), f); k = np.zeros(len(p), np.int32)
    live = np.ones(len(p), bool)
    for _ in range(8):
        rho2 = z[:, 0] * z[:, 0] + z[:, 1] * z[:, 1] + f(1e-30)
        r2n = z[:, 2] * z[:, 2] + rho2
        r2 = np.where(live, r2n, r2)
        live = live & ~(r2n > f(4.0))
        if not live.any(): break
        with np.errstate(all="ignore"):
            r = np.sqrt(r2n); q = f(1) / np.sqrt(rho2); rho = rho2 * q
            drn = f(8) * (r2n * r2n * r2n * r) * dr + f(1)
            A, B, C, D = z[:, 2], rho, z[:, 0] * q, z[:, 1] * q
            for _s in range(3):
                A, B = A * A - B * B, f(2) * A * B
                C, D = C * C - D * D, f(2) * C * D
            zn = np.stack([B * C + p[:, 0], B * D + p[:, 1], A + p[:, 2]], -1).astype(f)
        z = np.where(live[:, None], zn, z); dr = np.where(live, drn, dr); k += live
    with np.errstate(all="ignore"):
        d = (np.log2(r2) * f(0.17328680) * np.sqrt(r2) / dr).astype(f)
    return d, k
W, H, N = 3840, 2160, 256
th = f(np.tan(f(1.5) / 2))
def tile_rays(x0, y0, w=16, h=32):
    xs, ys = np.meshgrid(np.arange(x0, x0 + w), np.arange(y0, y0 + h))
    tx = (xs.ravel() + 0.5) / W; ty = (ys.ravel() + 0.5) / H
    d = np.stack([(tx * 2 - 1) * (W / H) * th, (ty * 2 - 1) * th, np.ones(tx.size)], -1).astype(f)
    d /= np.linalg.norm(d, axis=-1, keepdims=True).astype(f)
    # wave-major order of the kernel: 8x8 pixel waves inside the tile
    order = np.argsort(((ys.ravel() - y0) // 8) * 2 * 64 + ((xs.ravel() - x0) // 8) * 64 + ((ys.ravel() - y0) % 8) * 8 + (xs.ravel() - x0) % 8, kind="stable")
    return d[order]
rng = np.random.default_rng(3)
tiles = [(int(x) // 16 * 16, int(y) // 32 * 32) for x, y in zip(rng.integers(1100, 2740, 40), rng.integers(400, 1760, 40))]
tot = {"used": 0.0, "a": 0.0, "b": 0.0, "c": 0.0, "steps_a": 0.0, "steps_b": 0.0}
for (x0, y0) in tiles:
    dirs = tile_rays(x0, y0); n = len(dirs)
    p = np.tile(np.array([0, 0, -2.5], f), (n, 1))
    ks = np.zeros((N, n), np.int32); alive = np.zeros((N, n), bool)
    live = np.ones(n, bool)
    for i in range(N):
        d, k = de(p)
        q = (p + (dirs * d[:, None]).astype(f)).astype(f)
        ks[i] = k; alive[i] = live
        live = live & ~(q == p).all(-1) & np.isfinite(q).all(-1)
        p = np.where(alive[i][:, None], q, p)
        if not live.any(): break
    # rays escaped to inf/nan are "settled" in the kernel too (position stops changing)
    tot["used"] += (ks * alive).sum()
    # (a) kernel: slots; stable compaction every 16 steps when it frees a wave
    def run(sort_by_k):
        slot_of = np.arange(n); issued = 0; wave_steps = 0
        for i in range(N):
            if i % 16 == 0 and i > 0:
                act = np.nonzero(alive[i])[0]
                if len(act) == 0: break
                used_waves = len(np.unique(slot_of[act] // 64)); needed = (len(act) + 63) // 64
                if sort_by_k:
                    key = ks[i - 1][act]
                    act = act[np.argsort(-key, kind="stable")]
                    slot_of[act] = np.arange(len(act))
                elif needed < used_waves:
                    act = act[np.argsort(slot_of[act], kind="stable")]
                    slot_of[act] = np.arange(len(act))
            act = np.nonzero(alive[i])[0]
            if len(act) == 0: break
            w = slot_of[act] // 64
            mx = np.zeros(8, np.int32); np.maximum.at(mx, w, ks[i][act] + 1)  # +1: the bail-out check of the round that ends it
            issued += mx.sum(); wave_steps += len(np.unique(w))
        return issued, wave_steps
    a, sa = run(False); b, sb = run(True)
    tot["a"] += a; tot["b"] += b; tot["steps_a"] += sa; tot["steps_b"] += sb
    c = 0
    for i in range(N):
        act = np.nonzero(alive[i])[0]
        if len(act) == 0: break
        kk = np.sort(ks[i][act] + 1)[::-1]
        c += kk[::64].sum()
    tot["c"] += c
    tot["used"] += alive.sum()  # the +1 checks
used = tot["used"] / 64
print("lane-rounds used / 64: %.0f" % used)
for name in "abc":
    print("(%s) wave-rounds issued %.0f  used/issued %.3f" % (name, tot[name], used / tot[name]))
print("wave-steps: kernel repack %.0f, sorted repack %.0f" % (tot["steps_a"], tot["steps_b"]))
