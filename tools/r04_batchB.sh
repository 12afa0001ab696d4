#!/bin/bash
# round 4, evidence B (final sources, counters collected): bench lines, the build x implementation table, one GPU standing in for the ranks, the present by parts, the fuzz log
export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
{
echo "bench.py --workload <w> --steps 6 --warmup 2 --no-cpu-baseline on the final sources (one JSON line each; the headline's default run is r04_bench_default.json):"
for w in "--workload c2" "--workload c3a" "--workload c4" "--workload c5 --steps 3 --warmup 1" "--workload c4 --stripe-of 8" "--workload c5 --stripe-of 8" "--strict" "--strict --workload c4 --steps 3 --warmup 1" "--strict --workload c5 --stripe-of 8 --steps 3 --warmup 1"; do
  python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline $w 2>/dev/null | tail -1
done
} > $O/bench_workloads.txt
python3 tools/time_all.py > $O/time_all.txt 2>&1
python3 tools/emulate_ranks.py > $O/shard_emulation.txt 2>&1
EMU_YIELD=1 python3 tools/emulate_ranks.py >> $O/shard_emulation.txt 2>&1
python3 tools/time_present.py > $O/present_by_parts.txt 2>&1
RM_BENCH_SHARE_GPU=1 python3 bench.py --gpus 4 --dof --check-frame --steps 16 --warmup 8 > $O/bench_ranks_sharing_dof.txt 2>&1
bash tools/fuzz.sh r04 > $O/fuzz_stdout.txt 2>&1
tail -c 600 $O/bench_default.json; cat $O/time_all.txt; tail -12 $O/shard_emulation.txt; tail -3 $O/bench_ranks_sharing_dof.txt | cut -c1-300; cat gpurun_out/r04_fuzz/log.txt gpurun_out/r04_fuzz/log_seeds.txt gpurun_out/r04_fuzz/log_gl.txt gpurun_out/r04_fuzz/log_abuse.txt
