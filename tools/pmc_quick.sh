#!/bin/bash
# one PMC pass, per-kernel averages: bash tools/pmc_quick.sh <tag> "<counters>" [bench args]
TAG=$1; CTRS=$2; shift 2
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "run", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "rm::" not in k: continue
        acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]; print(f"   {c:26s} n={len(v):3d} mean {sum(v)/len(v):.5g}")
PY
