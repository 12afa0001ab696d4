#!/bin/bash
# round 4, evidence A (final sources): counters for every workload, the phase table of the headline kernel and of C4 / C5's stripes
export TMPDIR=/tmp
bash tools/profile_all.sh r04 > gpurun_out/r04_profile_all.log 2>&1
bash tools/phase_cost.sh r04_phase > gpurun_out/r04_phase.txt 2>&1
python3 tools/phase_table.py c4 > gpurun_out/r04_phase_c4.txt 2>&1
python3 tools/phase_table.py c5s > gpurun_out/r04_phase_c5s.txt 2>&1
tail -40 gpurun_out/r04_profile_all.log | cut -c1-200; cat gpurun_out/r04_phase.txt gpurun_out/r04_phase_c4.txt gpurun_out/r04_phase_c5s.txt
