#!/bin/bash
# the randomised tests of the GPU suite at 50-100x their default counts, for the record
mkdir -p gpurun_out/r2fuzz
{
date
RM_RANDOM_JOBS=20000 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "random_jobs_strict" 2>&1 | grep "random jobs\|passed\|failed"
RM_RANDOM_SCENES=20000 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_scenes_probes" 2>&1 | tail -1
RM_RANDOM_JOBS2=20000 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_jobs_partitions" 2>&1 | tail -1
for s in 11 12 13; do SEED=$s timeout 300 python3 tools/dbg/abuse_fuzz.py 10000 2>&1 | tail -1; done
date
} | tee gpurun_out/r2fuzz/log.txt
# other streams: every randomised test of the suite under three more seeds, at 10x the default counts
for seed in 1 2 3; do
  RM_RANDOM_SEED=$seed RM_RANDOM_JOBS=3000 RM_RANDOM_SCENES=2000 RM_RANDOM_JOBS2=1500 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random" 2>&1 | tail -2
done | tee gpurun_out/r2fuzz/log_seeds.txt
# the GL-stack arithmetic against the oracle in the same arithmetic, 20 000 jobs
RM_RANDOM_SEED=7 RM_RANDOM_GL_JOBS=20000 timeout 1200 python -m pytest tests/test_gpu_reference_bits.py -m gpu -q -x -k random_jobs_equal 2>&1 | tail -1 | tee gpurun_out/r2fuzz/log_gl.txt
