#!/bin/bash
# kernel-trace stats only: bash tools/prof_quick.sh <tag> [bench args]
TAG=${1:-q}; shift
OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_trace.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print(f"{row.get('Name','')[:100]:100s} calls {row.get('Calls'):>4} avg_us {float(row.get('AverageNs'))/1e3:10.1f} pct {row.get('Percentage')}")
PY
tail -1 $OUT/bench_trace.log | cut -c1-200
