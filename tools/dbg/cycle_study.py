#!/usr/bin/env python3
"""Does the march of a ray that sits on the Mandelbulb's surface enter a short cycle?  (CPU, oracle's strict sdf.)
The march is the iteration p <- p + dir * sdf(p) of a deterministic function on floats: once a position repeats, the rest
of the 256 steps is periodic and the end position follows by modular arithmetic -- exactly.  Prints, for rays of the
headline camera that hit the fractal: the step of the first repeat, the cycle length, and the steps a detector would save."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from raymarching_engine_amd import scene as S
sc = S.Mandelbulb()
W, H, N = 3840, 2160, 256
rng = np.random.default_rng(1)
n = 3000
px = rng.integers(1200, 2640, n); py = rng.integers(500, 1660, n)
th = np.float32(np.tan(np.float32(1.5) / 2))
tx = (px + 0.5) / W; ty = (py + 0.5) / H
d = np.stack([(tx * 2 - 1) * (W / H) * th, (ty * 2 - 1) * th, np.ones(n)], -1).astype(np.float32)
d /= np.linalg.norm(d, axis=-1, keepdims=True).astype(np.float32)
p = np.tile(np.array([0, 0, -2.5], np.float32), (n, 1))
hist = [p.copy()]
for i in range(N):
    dist = O.eval_sdf(sc, p).astype(np.float32)
    p = (p + (d * dist[:, None]).astype(np.float32)).astype(np.float32)
    hist.append(p.copy())
hist = np.stack(hist)  # N+1, n, 3
hit = np.isfinite(hist[-1]).all(-1) & (np.abs(hist[-1]).max(-1) < 2.0)
print("rays", n, "on the fractal at the end", int(hit.sum()))
first_rep, lam = [], []
for r in np.nonzero(hit)[0]:
    seen = {}
    fr = None
    for i in range(N + 1):
        key = hist[i, r].tobytes()
        if key in seen:
            fr = (i, i - seen[key]); break
        seen[key] = i
    if fr: first_rep.append(fr[0]); lam.append(fr[1])
    else: first_rep.append(N + 1); lam.append(0)
first_rep = np.array(first_rep); lam = np.array(lam)
cyc = lam > 0
print("rays whose position repeats within 256 steps: %.3f" % cyc.mean())
print("step of the first repeat: median %d, p25 %d, p75 %d, p90 %d" % tuple(np.percentile(first_rep[cyc], [50, 25, 75, 90])))
print("cycle length: 1: %.3f  2: %.3f  3-4: %.3f  5-8: %.3f  9-16: %.3f  >16: %.3f" % tuple(((lam[cyc] >= a) & (lam[cyc] <= b)).mean() for a, b in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 999))))
saved = np.where(cyc, N - first_rep, 0)
print("steps after the first repeat (what an ideal detector saves), mean over the hit rays: %.1f of %d" % (saved.mean(), N))
