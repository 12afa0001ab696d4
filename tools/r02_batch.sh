#!/bin/bash
# sample batches: the GPU suite, per-rank emulation over yield interval x batches in flight
mkdir -p gpurun_out/r2q
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r2q/tests.txt 2>&1; tail -5 gpurun_out/r2q/tests.txt
for cfg in "8 1" "8 2" "8 3" "4 2" "4 3" "4 4" "2 4"; do
  set -- $cfg
  echo "== yield $1 depth $2"; EMU_YIELD=$1 EMU_DEPTH=$2 python3 tools/emulate_ranks.py 2>&1 | grep "^N=[48]"
done > gpurun_out/r2q/emu_sweep.txt 2>&1
cat gpurun_out/r2q/emu_sweep.txt
