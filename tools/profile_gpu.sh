#!/bin/bash
# rocprofv3 profile of a bench command on a GPU box; summaries land in gpurun_out/prof_<tag>/
#   bash tools/profile_gpu.sh <tag> "<bench args of the kernel-trace run>" "<bench args of each PMC pass>"
# The kernel-trace run and every counter group are separate rocprofv3 invocations (--pmc passes carry --kernel-trace only).
set -u
TAG=${1:-r01}
ARGS=${2:---steps 30 --warmup 5 --repeats 1 --no-cpu-baseline}
PMC_ARGS=${3:---steps 2 --warmup 1 --repeats 1 --no-cpu-baseline}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 bench.py $PMC_ARGS > $OUT/pmc_$name.log 2>&1
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
tail -1 $OUT/bench_trace.log | cut -c1-400 >> $OUT/summary.txt
cat $OUT/summary.txt
