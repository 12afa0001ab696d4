#!/bin/bash
# round 4, second GPU batch: the GPU suite with the parity build's exits and the adversarial far-field tests, asm schedules of the
# power-8 round, how much of the table fold is the LDS, phase tables of C4 / C5's stripes, the build x implementation table
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
bash tools/qvariants.sh base asm1 asm2 base > $O/qv.txt 2>&1
python3 tools/r03_table.py default tools/_exp_halflds.so > $O/table.txt 2>&1
python3 tools/r03_phase.py c4 > $O/phase_c4.txt 2>&1
python3 tools/r03_phase.py c5s > $O/phase_c5s.txt 2>&1
python3 tools/time_all.py > $O/time_all.txt 2>&1
tail -25 $O/pytest.log; cat $O/qv.txt $O/phase_c4.txt $O/phase_c5s.txt $O/time_all.txt; grep "c4 full\|c5 full\|==" $O/table.txt
