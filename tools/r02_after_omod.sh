#!/bin/bash
# after the output-modifier round: profile, per-rank emulation, image statistics fast vs strict, RCCL path at one rank
mkdir -p gpurun_out/r2p
bash tools/profile_gpu.sh r02d > /dev/null 2>&1
cp gpurun_out/prof_r02d/summary.txt gpurun_out/r2p/summary_r02d.txt
python3 tools/emulate_ranks.py > gpurun_out/r2p/shard_emulation.txt 2>&1
SPP=128 python3 tools/normal_study.py default > gpurun_out/r2p/fast_vs_strict.txt 2>&1
RM_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline > gpurun_out/r2p/force_dist.log 2>&1
python3 bench.py > gpurun_out/r2p/bench.log 2>&1
tail -30 gpurun_out/r2p/shard_emulation.txt; cat gpurun_out/r2p/fast_vs_strict.txt; tail -2 gpurun_out/r2p/force_dist.log; tail -1 gpurun_out/r2p/bench.log; head -8 gpurun_out/r2p/summary_r02d.txt
