#!/usr/bin/env python3
"""Does the fast build estimate what the strict build estimates, on RANDOM jobs?  For each job (random scene of every kind,
camera, lights, bounces; full mode): the image mean of 32 strict samples, of 32 fast samples with the same random stream,
and of 32 strict samples with ANOTHER stream (the Monte-Carlo yardstick).  Flags jobs whose fast / strict difference is
more than 4x the strict / strict one.  python tools/dbg/fast_vs_strict_fuzz.py [jobs]"""
import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
spec = importlib.util.spec_from_file_location("tg", os.path.join(ROOT, "tests", "test_gpu_parity.py")); tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native
n_jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 120
ctx = native.Context(0)
rng = np.random.default_rng(2026)
SPP = 32
all_noise = GC.halton_pairs(2 * SPP)
flagged = 0; ratios = []
for it in range(n_jobs):
    sc, pos = tg._random_scene(rng)
    w, h = 96, 64
    lo = int(os.environ.get("MIN_STEPS", "24"))
    counts = tuple(int(c) for c in rng.integers(lo, lo + 72, size=rng.integers(1, 4)))
    lights = [J.point_light(tuple(rng.uniform(-4, 4, 3)), size=float(rng.choice([0.0, 0.3]))) for _ in range(int(rng.integers(0, 3)))]
    cam = ("perspective", "perspective", "orthographic", "panoramic")[rng.integers(0, 4)]
    schema = J.make_schema(sc, w, h, counts=counts, render_mode="full", position=tuple(np.array(pos) + rng.uniform(-0.2, 0.2, 3)), rotation=GC.ROT if rng.random() < 0.5 else None,
                           camera=cam, fov=float(rng.uniform(0.8, 1.8)) if cam != "orthographic" else float(rng.uniform(2.0, 5.0)), lights=lights, fog_density=float(rng.choice([0.0, 0.0, 0.1])))
    def mean(flags, noises):
        img = tg.render_gpu(ctx, sc, schema, noises, flags)[0][..., :3] / len(noises)
        fin = np.isfinite(img).all(-1)
        return img, fin
    s1, f1 = mean(tg.STRICT, all_noise[:SPP]); fa, f2 = mean(tg.FAST, all_noise[:SPP]); s2, f3 = mean(tg.STRICT, all_noise[SPP:])
    fin = f1 & f2 & f3
    if fin.mean() < 0.2: continue
    m = float(s1[fin].mean())
    d_fast = float(abs(fa[fin].mean() - m)); d_mc = float(abs(s2[fin].mean() - m))
    rms_fast = float(np.sqrt(((fa[fin] - s1[fin]) ** 2).mean())); rms_mc = float(np.sqrt(((s2[fin] - s1[fin]) ** 2).mean()))
    nonfinite_same = float((f1 == f2).mean())
    ratios.append(rms_fast / max(rms_mc, 1e-12))
    bad = (d_fast > 4 * d_mc + 0.002 * abs(m) + 1e-6) or nonfinite_same < 0.98
    if bad:
        flagged += 1
        print(f"job {it}: {type(sc).__name__} params {[round(float(x), 3) for x in sc.params()][:7]} counts {counts} cam {cam} lights {len(lights)}: mean {m:.5f} fast-strict {d_fast:.2e} strict-strict {d_mc:.2e} rms {rms_fast:.2e} / {rms_mc:.2e} same finiteness {nonfinite_same:.3f}")
print(f"{n_jobs} jobs, {flagged} flagged; rms(fast - strict) / rms(strict' - strict): median {np.median(ratios):.3f}, max {np.max(ratios):.3f}")
