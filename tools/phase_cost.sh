#!/bin/bash
# cumulative cost of the pixel kernel's phases on the headline frame (diagnostic builds from tools/phase_cost.py)
OUT=$PWD/gpurun_out/${1:-phase}; mkdir -p $OUT; export TMPDIR=/tmp
for n in 1 2 3 4 5 0; do
  if [ $n = 0 ]; then unset RM_LIB; else export RM_LIB=$PWD/tools/_exp_stop$n.so; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/stop$n -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/stop$n.log 2>&1
  # the instruction mix of the same build, a pass of its own
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --kernel-trace --output-format csv -d $OUT/stop$n/mix -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/stop${n}_mix.log 2>&1
  python3 - $OUT/stop$n $n <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list); dur = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rm_pixel_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rm_pixel_kernel" in row["Kernel_Name"]:
            dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
dur = sorted(dur)
print("stop", sys.argv[2], "kernel ms (median, profiled)", "%.3f" % dur[len(dur) // 2], " ".join(f"{k} {sum(v) / len(v):.4g}" for k, v in sorted(acc.items())))
PY
done
