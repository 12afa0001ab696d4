#!/bin/bash
# Round 6's evidence refresh on the final sources, ONE call on a GPU box:   gpurun -- 'bash tools/round6_evidence.sh'
# bench.py measures its own hardware counters since round 6 (live_counters), so the per-workload profile_all.sh passes of rounds 2-5 are
# gone; what is left: the default line (with --dump-counters: the file a run that cannot measure replays), the rocprofv3 --kernel-trace
# --stats summary of the SAME command (profile_gpu.sh: its average kernel duration has to agree with roofline.kernel_ms), one bench line per
# other workload / build to hold the `workloads` legs of the default line against, the implementation table, one GPU standing in for the ranks
# of 2 / 4 / 8, and the ranks-sharing-one-GPU line of the sharded path (frame check on by default).  Everything lands under gpurun_out/r06/.
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
python3 bench.py --dump-counters $O/counters.json > $O/bench_default.json 2> $O/bench_default.err
bash tools/profile_gpu.sh r06_c3b "--steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-workloads" "--steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-workloads" > $O/profile_c3b.log 2>&1
{
echo "bench.py <flags> --steps 6 --warmup 2 --no-cpu-baseline --no-live-counters on round 6's final sources, one JSON line each (the default run's line, with these as its"
echo "\`workloads\` legs, is r06_bench_default.json):"
for w in "--workload c2 --steps 20" "--workload c3a" "--workload c4" "--workload c5 --steps 3 --warmup 1" "--strict" "--gl-stack 2 --steps 4" "--gl-stack 1 --steps 4" "--workload c4 --stripe-of 8" "--workload c5 --stripe-of 8" "--strict --workload c4 --steps 3 --warmup 1"; do
  timeout 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-live-counters $w 2>/dev/null | tail -1
done
} > $O/bench_workloads.txt
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -ffp-contract=off -o fold_rate fold_rate.hip > /dev/null 2>&1); timeout 120 tools/ubench/fold_rate > $O/fold_rate.txt 2>&1
timeout 600 python3 tools/time_all.py > $O/time_all.txt 2>&1
timeout 600 python3 tools/emulate_ranks.py > $O/shard_emulation.txt 2>&1
EMU_YIELD=1 timeout 600 python3 tools/emulate_ranks.py >> $O/shard_emulation.txt 2>&1
RM_BENCH_SHARE_GPU=1 RM_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 4 --steps 16 --warmup 8 --no-cpu-baseline > $O/bench_ranks_sharing.txt 2>&1
RM_BENCH_SHARE_GPU=1 RM_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 4 --steps 16 --warmup 8 --no-cpu-baseline --dof > $O/bench_ranks_sharing_dof.txt 2>&1
RM_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline > $O/bench_rccl_one_rank.txt 2>&1
tail -c 400 $O/bench_default.json; echo; cat $O/time_all.txt | tail -20; tail -12 $O/shard_emulation.txt
