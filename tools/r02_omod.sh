#!/bin/bash
# omod experiment: probe, timings of the variants, the Mandelbulb parity tests per variant
mkdir -p gpurun_out/r2o
./tools/ubench/omod_probe > gpurun_out/r2o/probe.txt 2>&1
bash tools/qvariants.sh base om1 om2 om3 om4 > gpurun_out/r2o/timing.txt 2>&1
for v in om1 om2 om4 om3; do
  RM_LIB=$PWD/tools/_exp_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "c3b or mandelbulb or fast_build or wavefront" > gpurun_out/r2o/tests_$v.txt 2>&1
  tail -3 gpurun_out/r2o/tests_$v.txt
done
cat gpurun_out/r2o/probe.txt gpurun_out/r2o/timing.txt
