#!/bin/bash
# one PMC pass over any script: bash tools/pmc_script.sh <tag> "<counters>" <script.py> [args]
TAG=$1; CTRS=$2; shift 2
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/run -- python3 "$@" > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "run", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "rm::" not in k: continue
        acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(os.path.join(sys.argv[1], "run", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "rm::" in k: dur[k[:60]].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e6)
for k in acc:
    print(k, "ms", ["%.2f" % x for x in dur.get(k, [])][:6])
    for c in sorted(acc[k]):
        v = acc[k][c]; print(f"   {c:26s} n={len(v):3d} mean {sum(v)/len(v):.5g}")
PY
