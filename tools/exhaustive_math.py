#!/usr/bin/env python3
"""Every one-argument function of the parity arithmetic on EVERY float: the kernels' compilation (rm_probe_math) against the oracle's.

  python tools/exhaustive_math.py [strict|gl] [functions...]      default: strict, all one-argument functions

All 2^32 bit patterns per function, in chunks; the oracle side runs on the host's cores (ctypes releases the GIL).  The two-argument
functions (pow, atan2, div, the pow pair) get SAMPLES x 2^24 random pairs of bit patterns plus pairs of ordinary size instead.
Prints one line per function: arguments compared, arguments that differ (NaN == NaN)."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402  (the checker; this is a measurement script, not the product)
from raymarching_engine_amd import native  # noqa: E402

ONE = ("sin", "cos", "log", "exp", "acos", "tan", "sincos_s", "sincos_c", "sqrt")
TWO = ("pow", "atan2", "div", "pow_pair_nm1", "pow_pair_n")
CHUNK = 1 << 24
SAMPLES = int(os.environ.get("SAMPLES", "24"))


def same(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("strict", "gl") else "strict"
    names = [a for a in sys.argv[1:] if a not in ("strict", "gl")] or list(ONE + TWO)
    ctx = native.Context(0)
    if mode == "gl":
        ctx.set_gl_stack(1)
        O.set_math_mode(O.MATH_SWIFTSHADER)
    workers = min(32, os.cpu_count() or 8)
    print(f"{mode} arithmetic, oracle on {workers} threads")
    pool = ThreadPoolExecutor(workers)
    for name in names:
        t0 = time.time()
        bad = total = 0
        first = []
        if name in ONE:
            chunks = ((np.arange(c * CHUNK, (c + 1) * CHUNK, dtype=np.uint64).astype(np.uint32).view(np.float32), None) for c in range(1 << 8))
        else:
            rng = np.random.default_rng(99)

            def pairs():
                for k in range(SAMPLES):
                    if k % 2 == 0:  # any two bit patterns
                        yield (rng.integers(0, 1 << 32, CHUNK, dtype=np.uint64).astype(np.uint32).view(np.float32),
                               rng.integers(0, 1 << 32, CHUNK, dtype=np.uint64).astype(np.uint32).view(np.float32))
                    else:  # numbers of a shader's size
                        a = (np.exp(rng.uniform(-12, 12, CHUNK)) * rng.choice([-1.0, 1.0], CHUNK)).astype(np.float32)
                        b = (rng.uniform(-70, 70, CHUNK) if name.startswith("pow") else np.exp(rng.uniform(-12, 12, CHUNK)) * rng.choice([-1.0, 1.0], CHUNK)).astype(np.float32)
                        yield a, b
            chunks = pairs()
        pending = []
        for a, b in chunks:
            parts = np.array_split(np.arange(a.size), workers)
            want = np.empty_like(a)

            def oracle_part(idx, a=a, b=b, want=want):
                want[idx] = O.math(name, a[idx], None if b is None else b[idx])
            futs = [pool.submit(oracle_part, idx) for idx in parts]
            got = ctx.probe_math(name, a, b)
            for f in futs:
                f.result()
            eq = same(got, want)
            total += a.size
            if not eq.all():
                i = np.flatnonzero(~eq)
                bad += i.size
                if len(first) < 4:
                    first += [f"f({a[j]!r}{'' if b is None else ', ' + repr(b[j])}) = {got[j]!r} vs {want[j]!r}" for j in i[:4 - len(first)]]
        print(f"{name:14s} {total:12d} arguments, {bad} differ  ({time.time() - t0:.0f} s)" + ("  e.g. " + "; ".join(first) if first else ""), flush=True)


if __name__ == "__main__":
    main()
