#!/bin/bash
# round 4, first GPU batch: GPU suite on the new sources, headline variants, the phase table, table kernels by waves per SIMD
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python3 tools/r03_jump.py default tools/_exp_r3.so tools/_exp_cxxround.so tools/_exp_shadefast.so > $O/jump.txt 2>&1
bash tools/qvariants.sh base r3 cxxround shadefast > $O/qv.txt 2>&1
bash tools/phase_cost.sh r4a/phase > $O/phase.txt 2>&1
python3 tools/r03_table.py default tools/_exp_tw6.so tools/_exp_tw5.so > $O/table.txt 2>&1
tail -3 $O/pytest.log; cat $O/qv.txt $O/phase.txt
