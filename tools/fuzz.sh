#!/bin/bash
# The randomised tests of the GPU suite at scale, for the record (profiles/<round>_fuzz_log.txt): bash tools/fuzz.sh <round>
#  * random jobs / scenes / partitions of the strict build against the oracle at 30-40x their default counts (since round 4 two tables in
#    ten carry kind rows, four in ten surfaces), and under three more seeds;
#  * the exact far-field exits: the adversarial rays of tests/test_gpu_far_field.py at 30x (RM_FAR_RAYS) under three seeds, both builds,
#    and the random far-field / row-culling / shadow-ray tests of tests/test_gpu_parity.py under six seeds;
#  * GL-stack jobs against the oracle in the same arithmetic; nasty inputs through the C ABI.
R=${1:-r04}; O=gpurun_out/${R}_fuzz; mkdir -p $O
{
date
RM_RANDOM_JOBS=12000 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "random_jobs_strict" 2>&1 | grep "random jobs\|passed\|failed"
RM_RANDOM_SCENES=12000 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_scenes_probes" 2>&1 | tail -1
RM_RANDOM_JOBS2=6000 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_jobs_partitions" 2>&1 | tail -1
for seed in 1 2 3; do echo "far field x30, seed $seed"; RM_RANDOM_SEED=$seed RM_FAR_RAYS=30 timeout 1500 python -m pytest tests/test_gpu_far_field.py -m gpu -q -x 2>&1 | tail -1; done
for seed in 1 2 3 4 5 6; do RM_RANDOM_SEED=$seed timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "far_jump or far_field or row_culling or shadow_rays" 2>&1 | tail -1; done
for b in fast strict; do echo "row culling of random tables of every kind (hard operators, mixed, mostly smooth), 200 more tables each, seed 9, $b"; RM_RANDOM_SEED=9 RM_CULL_TABLES=200 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "row_culling_is_exact and $b" 2>&1 | tail -1; done
for seed in 7 8; do for b in fast strict; do echo "row culling of smooth sphere tables, 300 more tables of 16..256 rows, seed $seed, $b"; RM_RANDOM_SEED=$seed RM_CULL_TABLES=300 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "row_culling_of_smooth and $b" 2>&1 | tail -1; done; done
date
} 2>&1 | tee $O/log.txt
for seed in 1 2 3; do
  RM_RANDOM_SEED=$seed RM_RANDOM_JOBS=2000 RM_RANDOM_SCENES=1500 RM_RANDOM_JOBS2=1000 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random" 2>&1 | tail -2
done | tee $O/log_seeds.txt
RM_RANDOM_SEED=9 RM_RANDOM_GL_JOBS=8000 timeout 1500 python -m pytest tests/test_gpu_reference_bits.py -m gpu -q -x -k random_jobs_equal 2>&1 | tail -1 | tee $O/log_gl.txt
for s in 41 42 43 44; do SEED=$s timeout 300 python3 tools/abuse_fuzz.py 4000 2>&1 | tail -1; done | tee $O/log_abuse.txt
