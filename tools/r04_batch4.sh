#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O
python3 tools/table_variants.py tools/_exp_nopark.so tools/_exp_park.so tools/_exp_tw4.so tools/_exp_nopark.so tools/_exp_park.so > $O/table3.txt 2>&1
grep "==\|c4 full pixel\|c4 shard/8 pixel\|c5 shard/8 pixel\|c5 full pixel\|csg_mixed\|blocks (192 rows, hard operators) 1080p pixel\|Error\|error" $O/table3.txt
python3 tools/time_variants.py tools/_exp_nopark.so tools/_exp_park.so tools/_exp_nopark.so tools/_exp_park.so
