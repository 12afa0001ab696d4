"""Condense rocprofv3 csv output (kernel stats + PMC passes) into a short text summary."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Name", "")[:90]
        print(f"{name:90s} calls {row.get('Calls')} avg_ns {row.get('AverageNs')} min {row.get('MinNs')} max {row.get('MaxNs')} pct {row.get('Percentage')}")
print("== PMC (per pixel-kernel dispatch, averaged) ==")
acc = defaultdict(list)
meta = {}
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "rm_pixel_kernel" not in k:
            continue
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        meta = {x: row.get(x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} n={len(v)} mean {sum(v)/len(v):.6g}")
print("meta", meta)
