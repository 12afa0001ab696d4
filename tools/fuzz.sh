#!/bin/bash
# Round 3: the randomised tests of the GPU suite at 30-60x their default counts with the round's additions in their streams --
# four tables in ten carry per-shape surfaces, the fast march jumps escaping rays -- plus other seeds and the far-jump probe at scale.
mkdir -p gpurun_out/r3fuzz
{
date
RM_RANDOM_JOBS=12000 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "random_jobs_strict" 2>&1 | grep "random jobs\|passed\|failed"
RM_RANDOM_SCENES=12000 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_scenes_probes" 2>&1 | tail -1
RM_RANDOM_JOBS2=6000 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_jobs_partitions" 2>&1 | tail -1
for seed in 1 2 3 4 5 6; do RM_RANDOM_SEED=$seed timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "far_jump_end_points or row_culling or shadow_rays" 2>&1 | tail -1; done
date
} | tee gpurun_out/r3fuzz/log.txt
for seed in 1 2 3; do
  RM_RANDOM_SEED=$seed RM_RANDOM_JOBS=2000 RM_RANDOM_SCENES=1500 RM_RANDOM_JOBS2=1000 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random" 2>&1 | tail -2
done | tee gpurun_out/r3fuzz/log_seeds.txt
RM_RANDOM_SEED=9 RM_RANDOM_GL_JOBS=8000 timeout 1500 python -m pytest tests/test_gpu_reference_bits.py -m gpu -q -x -k random_jobs_equal 2>&1 | tail -1 | tee gpurun_out/r3fuzz/log_gl.txt
for s in 31 32 33 34; do SEED=$s timeout 300 python3 tools/dbg/abuse_fuzz.py 4000 2>&1 | tail -1; done | tee gpurun_out/r3fuzz/log_abuse.txt
