#!/bin/bash
# tools/one_kernel.sh [-DOK_KIND=9 -DOK_SHAPE=1 ... other -D flags]: registers, spills and instruction counts of ONE pixel kernel (no GPU needed)
cd "$(dirname "$0")/.."
out=/tmp/one_kernel_$$
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -ffp-contract=off --cuda-device-only -S \
  -Rpass-analysis=kernel-resource-usage "$@" tools/one_kernel.hip -o $out.s 2> $out.txt
grep -E "error|SGPRs:|VGPRs:|ScratchSize|Spill|Occupancy|LDS Size" $out.txt | sed 's/.*remark: [^ ]* //'
echo "v_readlane $(grep -c v_readlane $out.s)  v_writelane $(grep -c v_writelane $out.s)  scratch_load $(grep -c scratch_load $out.s)  scratch_store $(grep -c scratch_store $out.s)  s_load $(grep -c 's_load' $out.s)  s_nop $(grep -c s_nop $out.s)  lines $(grep -cE '^\s+[vs]_|^\s+ds_|^\s+global_|^\s+scratch_' $out.s)"
echo "asm: $out.s"
