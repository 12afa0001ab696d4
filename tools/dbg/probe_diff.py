import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
spec = importlib.util.spec_from_file_location("tg", os.path.join(ROOT, "tests", "test_gpu_parity.py")); tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
from oracle import oracle as O
from raymarching_engine_amd import abi, native
ctx = native.Context(0)
rng = np.random.default_rng(31337)
special = np.array([0.0, -0.0, 1.0, -1.0, 1e-20, 1e20, 3e38, np.inf, -np.inf, np.nan], np.float32)
shown = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    sc, pos = tg._random_scene(rng)
    h = ctx.create_scene(sc)
    pts = rng.normal(scale=float(rng.choice([0.5, 2.0, 10.0])), size=(300, 3)).astype(np.float32)
    pts[:20] = rng.choice(special, size=(20, 3))
    pts[20:26] = np.eye(3, dtype=np.float32).repeat(2, 0) * np.float32(rng.uniform(0.1, 3.0))
    for name, what, fn, par in (("sdf", abi.RM_PROBE_SDF, O.eval_sdf, None), ("normal", abi.RM_PROBE_NORMAL, O.normal, 1e-5), ("material", abi.RM_PROBE_MATERIAL, O.material, None)):
        got = ctx.probe(h, what, pts, par or 0.0); want = fn(sc, pts, par) if par else fn(sc, pts)
        eq = tg.same_bits(got, want); eq = eq if eq.ndim == 1 else eq.all(1)
        if not eq.all() and shown < 12:
            for i in np.nonzero(~eq)[0][:4]:
                print(it, type(sc).__name__, name, "p", pts[i], "gpu", got[i], "oracle", want[i]); shown += 1
    # consume the same random numbers as the test
    org = rng.normal(scale=0.3, size=(120, 3)); dirs = rng.normal(size=(120, 3)); rng.choice([0.0, 1.0, 7.0, 33.0, 64.0, 12.5])
    h.destroy()
    rng.integers(3, 70); rng.integers(3, 50); rng.integers(0, 3); rng.random(); rng.uniform(-3, 3, 3); rng.uniform(0.3, 2.5); rng.choice([0.0, 0.1]); rng.uniform(0.5, 5.0); rng.random(); rng.random()
