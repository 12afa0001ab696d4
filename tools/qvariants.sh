#!/bin/bash
# headline timing of experiment libraries: bash tools/qvariants.sh name1 name2 ...   ("base" = the product library)
for v in "$@"; do
  if [ $v = base ]; then unset RM_LIB; else export RM_LIB=$PWD/tools/_exp_$v.so; fi
  echo "== $v"; bash tools/qbench.sh 2
done
