#!/bin/bash
# quick headline timing: N bench runs, kernel_ms / ms_per_step / frac on one line each.  bash tools/qbench.sh [runs] [bench args]
N=${1:-2}; shift
for i in $(seq $N); do timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step %.4f kernel_ms %.4f frac %.4f' % (d['ms_per_step'], r['kernel_ms'], r['frac']))"; done
