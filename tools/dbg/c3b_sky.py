import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_cases as GC
from oracle import oracle as O
from raymarching_engine_amd import abi, job as J, native, scene as S
import test_gpu_parity as T
ctx = native.Context(0)
O.set_tan_mode(O.TAN_PORTABLE)
sc, schema = T._c3b()
for crop, (x0, y0) in T.C3B_CROPS.items():
    w, h = 128, 32
    noises = GC.halton_pairs(1)
    want = T.render_oracle(sc, schema, noises, rows=(y0, h), tile=(x0, y0, w, h))
    for flags in (abi.RM_RENDER_STRICT, abi.RM_RENDER_FAST):
        got = T.render_gpu(ctx, sc, schema, noises, flags, rows=(y0, h), tile=abi.RmRect(x0, y0, w, h))
        depth = want[2][:, x0:x0 + w, 3]
        g, wv = got[0][:, x0:x0 + w], want[0][:, x0:x0 + w]
        d = T.rel_diff(wv, g).max(-1)
        sky = depth > 1e5
        bad = sky & (d > 2e-7)
        print(crop, flags, "sky", sky.mean(), "bad sky", bad.sum(), "quantiles", [float(np.quantile(d[sky], q)) for q in (0.9, 0.99, 0.999, 1.0)])
        ys, xs = np.where(bad)
        for k in range(min(6, len(ys))):
            print("   ", ys[k], xs[k], "want", wv[ys[k], xs[k]], "got", g[ys[k], xs[k]], "depth w/g", depth[ys[k], xs[k]], got[2][ys[k], x0 + xs[k], 3])
