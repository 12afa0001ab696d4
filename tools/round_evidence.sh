#!/bin/bash
# The round's ONE evidence refresh, on the final sources.  Two calls on a GPU box with `tools/collect_profiles.sh <round>` in between
# (the bench lines of part B replay the counters part A measured):
#   gpurun -- 'bash tools/round_evidence.sh r04 A'   then, in the build container,   bash tools/collect_profiles.sh r04
#   gpurun -- 'bash tools/round_evidence.sh r04 B'
# A: counters of every BASELINE workload (profile_all.sh), the phase table of the headline kernel with its instruction mix
#    (phase_cost.sh; needs the diagnostic builds of `python tools/phase_cost.py`), the phase tables of C4 and a C5 stripe.
# B: bench.py's default line and one line per other workload (strict ones included), the build x implementation table, one GPU
#    standing in for the ranks of 2 / 4 / 8, the present pass by parts, the ranks-sharing-one-GPU bench lines, the fuzz log.
# Everything lands under gpurun_out/; what is kept goes to profiles/<round>_* (profiles/README.md says which file is what).
export TMPDIR=/tmp
R=${1:?round tag, e.g. r04}; PART=${2:?A or B}
if [ $PART = A ]; then
  bash tools/profile_all.sh $R > gpurun_out/${R}_profile_all.log 2>&1
  bash tools/phase_cost.sh ${R}_phase > gpurun_out/${R}_phase.txt 2>&1
  python3 tools/phase_table.py c4 > gpurun_out/${R}_phase_c4.txt 2>&1
  python3 tools/phase_table.py c5s > gpurun_out/${R}_phase_c5s.txt 2>&1
  cat gpurun_out/${R}_phase.txt gpurun_out/${R}_phase_c4.txt gpurun_out/${R}_phase_c5s.txt
else
  O=gpurun_out/${R}b; mkdir -p $O
  python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
  {
  echo "bench.py --workload <w> --steps 6 --warmup 2 --no-cpu-baseline on the final sources (one JSON line each; the headline's default run is ${R}_bench_default.json):"
  for w in "--workload c2" "--workload c3a" "--workload c4" "--workload c5 --steps 3 --warmup 1" "--workload c4 --stripe-of 8" "--workload c5 --stripe-of 8" "--strict" "--strict --workload c4 --steps 3 --warmup 1" "--strict --workload c5 --stripe-of 8 --steps 3 --warmup 1"; do
    python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline $w 2>/dev/null | tail -1
  done
  } > $O/bench_workloads.txt
  python3 tools/time_all.py > $O/time_all.txt 2>&1
  python3 tools/emulate_ranks.py > $O/shard_emulation.txt 2>&1
  EMU_YIELD=1 python3 tools/emulate_ranks.py >> $O/shard_emulation.txt 2>&1
  python3 tools/time_present.py > $O/present_by_parts.txt 2>&1
  SPP=32 python3 tools/normal_study.py default > $O/normal_study.txt 2>&1
  python3 tools/small_kernels_probe.py > $O/small_kernels.txt 2>&1
  RM_BENCH_SHARE_GPU=1 RM_BENCH_BACKEND=gloo python3 bench.py --gpus 4 --steps 16 --warmup 8 --no-cpu-baseline --check-frame --dof > $O/bench_ranks_sharing_dof.txt 2>&1
  RM_BENCH_SHARE_GPU=1 RM_BENCH_BACKEND=gloo python3 bench.py --gpus 4 --steps 16 --warmup 8 --no-cpu-baseline --check-frame > $O/bench_ranks_sharing.txt 2>&1
  bash tools/fuzz.sh $R > $O/fuzz_stdout.txt 2>&1
  tail -c 600 $O/bench_default.json; cat $O/time_all.txt; tail -12 $O/shard_emulation.txt
fi
