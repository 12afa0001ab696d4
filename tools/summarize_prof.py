"""Condense rocprofv3 csv output (kernel stats + PMC passes) into a short text summary.
Per kernel: the dispatch count n, the mean per dispatch and the SUM over the run's dispatches (a pipeline of several kernels
per frame is priced per frame by the sums: tools/update_counters.py)."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Name", "")[:90]
        print(f"{name:90s} calls {row.get('Calls'):>4} avg_ns {float(row.get('AverageNs')):12.0f} min {row.get('MinNs'):>9} max {row.get('MaxNs'):>9} pct {row.get('Percentage')}")
print("== PMC (per dispatch of each kernel, averaged; separate rocprofv3 --pmc passes) ==")
acc = defaultdict(lambda: defaultdict(list))
meta = {}
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "rm::" not in k:
            continue
        k = k[:70]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        meta[k] = {x: row.get(x) for x in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
for k in acc:
    print(k, meta[k])
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:26s} n={len(v):3d} mean {sum(v)/len(v):.6g} sum {sum(v):.6g}")
