#!/bin/bash
# sample batches: tests, per-rank emulation with and without them, the RCCL path at one rank
mkdir -p gpurun_out/r2q
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "batches or in_flight" > gpurun_out/r2q/tests.txt 2>&1; tail -5 gpurun_out/r2q/tests.txt
EMU_YIELD=8 python3 tools/emulate_ranks.py > gpurun_out/r2q/emu_yield8.txt 2>&1; tail -5 gpurun_out/r2q/emu_yield8.txt
EMU_YIELD=4 python3 tools/emulate_ranks.py > gpurun_out/r2q/emu_yield4.txt 2>&1; tail -4 gpurun_out/r2q/emu_yield4.txt
EMU_YIELD=1 python3 tools/emulate_ranks.py > gpurun_out/r2q/emu_yield1.txt 2>&1; tail -4 gpurun_out/r2q/emu_yield1.txt
RM_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline --yield-interval 8 > gpurun_out/r2q/force_dist_y8.log 2>&1; tail -1 gpurun_out/r2q/force_dist_y8.log | cut -c1-400
python3 bench.py --no-cpu-baseline | cut -c1-300
