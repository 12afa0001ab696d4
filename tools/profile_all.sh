#!/bin/bash
# Counter evidence for every BASELINE configuration, ONCE per round, on the round's final sources.  On a GPU box:
#   bash tools/profile_all.sh <round, e.g. r04> [workloads...]     default: all of them
# Each: kernel trace (--stats) + six PMC passes of the same bench command, summary -> gpurun_out/prof_<round>_<name>/summary.txt.
# Afterwards, in the build container: tools/collect_profiles.sh <round> copies the summaries to profiles/ and fills profiles/<round>_counters.json.
set -u
R=${1:?round tag, e.g. r04}; shift
ALL="c3b c3a c2 c4_mk c4_wf c4shard c5shard_wf c5shard_mk c5_mk c5_wf c3b_strict c4_strict"
for w in ${@:-$ALL}; do
  case $w in
    c3b)        T="--steps 30 --warmup 5";               P="--steps 2 --warmup 1";;
    c3a)        T="--workload c3a --steps 30 --warmup 5"; P="--workload c3a --steps 2 --warmup 1";;
    c2)         T="--workload c2 --steps 100 --warmup 10"; P="--workload c2 --steps 2 --warmup 1";;
    c4_mk)      T="--workload c4 --megakernel --steps 6 --warmup 2"; P="--workload c4 --megakernel --steps 2 --warmup 1";;
    c4_wf)      T="--workload c4 --wavefront --steps 6 --warmup 2";  P="--workload c4 --wavefront --steps 2 --warmup 1";;
    c4shard)    T="--workload c4 --stripe-of 8 --steps 10 --warmup 2"; P="--workload c4 --stripe-of 8 --steps 2 --warmup 1";;
    c5shard_wf) T="--workload c5 --stripe-of 8 --wavefront --steps 4 --warmup 1";  P="--workload c5 --stripe-of 8 --wavefront --steps 2 --warmup 1";;
    c5shard_mk) T="--workload c5 --stripe-of 8 --megakernel --steps 4 --warmup 1"; P="--workload c5 --stripe-of 8 --megakernel --steps 2 --warmup 1";;
    c5_mk)      T="--workload c5 --megakernel --steps 2 --warmup 1"; P="--workload c5 --megakernel --steps 2 --warmup 1";;
    c5_wf)      T="--workload c5 --wavefront --steps 2 --warmup 1";  P="--workload c5 --wavefront --steps 2 --warmup 1";;
    c3b_strict) T="--strict --steps 4 --warmup 1"; P="--strict --steps 2 --warmup 1";;
    c4_strict)  T="--workload c4 --strict --steps 2 --warmup 1"; P="--workload c4 --strict --steps 2 --warmup 1";;
  esac
  echo "######## $w"
  bash tools/profile_gpu.sh ${R}_$w "$T --repeats 1 --no-cpu-baseline" "$P --repeats 1 --no-cpu-baseline" > gpurun_out/prof_${R}_$w.log 2>&1
  tail -3 gpurun_out/prof_${R}_$w/summary.txt | cut -c1-300
done
