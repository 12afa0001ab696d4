"""Microbenchmark of the distance estimator alone: 8M points that all need the deep (8-round) evaluation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raymarching_engine_amd import abi, native, scene as S
ctx = native.Context(0)
sc = S.Mandelbulb(); h = ctx.create_scene(sc)
rng = np.random.default_rng(0)
n = 1 << 23
p = rng.normal(size=(n, 3)).astype(np.float32); p /= np.linalg.norm(p, axis=1, keepdims=True); p *= 0.35  # inside the set: never bails out
for flags in (1, 0):
    d = ctx.probe(h, abi.RM_PROBE_SDF, p, 0.0, flags)
    print("flags", flags, "mean d", float(np.nanmean(d)))
