#!/bin/bash
# C4 / C5 through bench.py: the full frame and one shard of the 8-way row split, pixel kernel and wavefront pipeline
OUT=$PWD/gpurun_out/${1:-r02_c45}; mkdir -p $OUT
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" > $OUT/$name.log 2>&1; tail -1 $OUT/$name.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline'] or {}
    print('$name', 'ms/step %.3f' % d['ms_per_step'], 'Mpix/s %.1f' % d['value'], 'kernel_ms %.3f' % r.get('kernel_ms',0), 'frac %.3f' % r.get('frac',0), d['config']['pipeline'])
except Exception as e: print('$name FAILED', e)
"; }
run c4_full_mk --workload c4 --steps 10 --warmup 3 --megakernel
run c4_full_wf --workload c4 --steps 10 --warmup 3 --wavefront
run c4_shard_mk --workload c4 --rows 1536:2048 --steps 20 --warmup 3 --megakernel
run c4_shard_wf --workload c4 --rows 1536:2048 --steps 20 --warmup 3 --wavefront
run c5_shard_mk --workload c5 --rows 3072:4096 --steps 5 --warmup 2 --megakernel
run c5_shard_wf --workload c5 --rows 3072:4096 --steps 5 --warmup 2 --wavefront
run c5_full_mk --workload c5 --steps 3 --warmup 1 --megakernel
run c5_full_wf --workload c5 --steps 3 --warmup 1 --wavefront
run c3a_mk --workload c3a --steps 20 --warmup 5
run c2_mk --workload c2 --steps 50 --warmup 5
