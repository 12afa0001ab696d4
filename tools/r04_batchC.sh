#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
RM_BENCH_SHARE_GPU=1 RM_BENCH_BACKEND=gloo python3 bench.py --gpus 4 --steps 16 --warmup 8 --no-cpu-baseline --check-frame --dof > $O/bench_ranks_sharing_dof.txt 2>&1
RM_BENCH_SHARE_GPU=1 RM_BENCH_BACKEND=gloo python3 bench.py --gpus 4 --steps 16 --warmup 8 --no-cpu-baseline --check-frame > $O/bench_ranks_sharing.txt 2>&1
bash tools/phase_cost.sh r04_phase2 > gpurun_out/r04_phase2.txt 2>&1
tail -1 $O/bench_ranks_sharing_dof.txt | cut -c1-600; tail -1 $O/bench_ranks_sharing.txt | cut -c1-300; cat gpurun_out/r04_phase2.txt
