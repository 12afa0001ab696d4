#!/bin/bash
# sweep bands x pass1 x pass2 blocks/CU on the headline config
for b in 1 2 3; do for p1 in 1 2 3 4; do for p2 in 1 2 3; do
  r=$(RM_WF_BANDS=$b RM_PASS1_BLOCKS_PER_CU=$p1 RM_PASS2_BLOCKS_PER_CU=$p2 timeout 60 python tools/time_variants.py default | sed 's/default //')
  echo "bands $b p1 $p1 p2 $p2 : $r"
done; done; done
