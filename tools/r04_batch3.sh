#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O
timeout 1800 python3 -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
