#!/bin/bash
# After tools/profile_all.sh <round> on a GPU box: keep the summaries under profiles/ and fill profiles/<round>_counters.json.
#   bash tools/collect_profiles.sh <round, e.g. r04> [names...]
# frames of a PMC pass (--steps 2 --warmup 1 --repeats 1): 1 + 2 + 2 + 3 = 8
set -u
R=${1:?round tag, e.g. r04}; shift
export RM_COUNTERS_ROUND=$R
ALL="c3b c3a c2 c4_mk c4_wf c4shard c5shard_wf c5shard_mk c5_mk c5_wf c3b_strict c4_strict"
for w in ${@:-$ALL}; do
  S=gpurun_out/prof_${R}_$w/summary.txt
  [ -f $S ] || { echo "no $S"; continue; }
  K=profiles/${R}_$w.txt
  cp $S $K
  case $w in
    c3b)        python3 tools/update_counters.py c3b_fast $S $K 8 $((3840*2160)) megakernel;;
    c3a)        python3 tools/update_counters.py c3a_fast $S $K 8 $((3840*2160)) megakernel;;
    c2)         python3 tools/update_counters.py c2_fast $S $K 8 $((1920*1080)) megakernel;;
    c4_mk)      python3 tools/update_counters.py c4_fast_megakernel $S $K 8 $((4096*4096)) megakernel;;
    c4_wf)      python3 tools/update_counters.py c4_fast_wavefront $S $K 8 $((4096*4096)) wavefront;;
    c4shard)    python3 tools/update_counters.py c4_shard_fast $S $K 8 $((4096*512)) megakernel;;
    c5shard_wf) python3 tools/update_counters.py c5_shard_fast_wavefront $S $K 8 $((8192*1024)) wavefront;;
    c5shard_mk) python3 tools/update_counters.py c5_shard_fast_megakernel $S $K 8 $((8192*1024)) megakernel;;
    c5_mk)      python3 tools/update_counters.py c5_fast_megakernel $S $K 8 $((8192*8192)) megakernel;;
    c5_wf)      python3 tools/update_counters.py c5_fast_wavefront $S $K 8 $((8192*8192)) wavefront;;
    c3b_strict) python3 tools/update_counters.py c3b_strict $S $K 8 $((3840*2160)) megakernel;;
    c4_strict)  python3 tools/update_counters.py c4_strict $S $K 8 $((4096*4096)) megakernel;;
  esac
done
