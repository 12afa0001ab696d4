#!/usr/bin/env python3
"""Instrumented algorithmic flops per pixel-sample of the bench workloads -> profiles/flops_per_pixel.json.

The numerator of bench.py's roofline.  The oracle built with -DOR_COUNT_FLOPS counts, under SURVEY.md 8(d)'s
convention, the flops the reference algorithm performs for every pixel (data-dependent branches included) for the
first sample of the job (randNoise = Halton index 0 = (1/2, 1/3)).  Rows are sampled with a fixed stride over the
WHOLE frame (stride 1 = every row); per-row sums are kept so that a reader can recompute any sub-sample and bench.py
can price a striped shard by the rows it holds.

    python tools/count_flops.py c3b --stride 1        # the headline frame, every row (about 10 min on 8 cores)
    python tools/count_flops.py c4 --stride 64
    python tools/count_flops.py c3b --stride 1 --pruned   # ... of the PRUNED algorithm (entry c3b_pruned): bench.py's frac_useful

Runs in the build container or anywhere else: it needs only the oracle (CPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "profiles", "flops_per_pixel.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--stride", type=int, default=16)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--pruned", action="store_true", help="do not count a march's steps behind its bitwise fixed point: the entry <workload>_pruned, the numerator of bench.py's frac_useful")
    args = ap.parse_args()

    import bench
    from oracle import oracle as O
    from raymarching_engine_amd import job as J

    O.build()
    O.set_tan_mode(O.TAN_PORTABLE)
    O.set_count_pruned(args.pruned)
    wl, sc, schema = bench.make_workload(args.workload)
    W, H = wl["width"], wl["height"]
    u = J.uniforms_from_schema(schema, (0.5, 1.0 / 3.0))
    threads = args.threads or O.host_cores()
    rows = list(range(args.stride // 2, H, args.stride))
    t0 = time.time()
    per_row = []
    # one single-row call per sampled row keeps the per-row sums; the calls run in a thread pool (ctypes drops the GIL)
    from concurrent.futures import ThreadPoolExecutor

    def one(r):
        f, _ = O.render_rows(sc, u, W, H, [r], threads=1, count_flops=True)
        return f

    chunk = max(threads * 8, 16)
    with ThreadPoolExecutor(max_workers=threads) as ex:
        for i in range(0, len(rows), chunk):
            per_row.extend(ex.map(one, rows[i:i + chunk]))
            print(f"{args.workload}: {min(i + chunk, len(rows))}/{len(rows)} rows, {time.time() - t0:.0f} s", flush=True)
    total = sum(per_row)
    entry = {
        "workload": wl["name"], "width": W, "height": H, "row_stride": args.stride, "first_row": rows[0], "rows": len(rows),
        "flops_per_pixel_sample": total / (len(rows) * W), "flops_per_row": per_row,
        "convention": "SURVEY.md 8(d): add/sub/mul/min/max/abs/compare/select 1, fma 2, transcendental or division 1, pow 2",
        "sample": "randNoise (1/2, 1/3), oracle/rm_oracle.c -DOR_COUNT_FLOPS, portable tangent", "seconds": round(time.time() - t0, 1),
    }
    if args.pruned:
        entry["pruned"] = "steps behind a march's bitwise fixed point are not counted (or_set_count_pruned): the arithmetic the marches need"
    data = {}
    if os.path.exists(OUT):
        data = json.load(open(OUT))
    data[args.workload + ("_pruned" if args.pruned else "")] = entry
    json.dump(data, open(OUT, "w"), indent=0)
    print(f"{args.workload}: {entry['flops_per_pixel_sample']:.1f} flop per pixel-sample over {len(rows)} rows (stride {args.stride})")


if __name__ == "__main__":
    main()
