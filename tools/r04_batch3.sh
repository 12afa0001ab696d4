#!/bin/bash
# round 4, third GPU batch: the GPU suite with the present in two halves / striped present, the present pass by parts, a bench line
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python3 tools/time_present.py > $O/time_present.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -18 $O/pytest.log; cat $O/time_present.txt; tail -c 1500 $O/bench.json
