#!/bin/bash
# debugging visit: run selected tests in separate processes, keep full logs
OUT=gpurun_out/${1:-dbg}; mkdir -p $OUT; export TMPDIR=/tmp
shift
i=0
for t in "$@"; do
  i=$((i+1))
  echo "== $t"
  timeout 300 python -m pytest "$t" -x -q -m gpu -p no:cacheprovider > $OUT/test_$i.log 2>&1
  grep -v '^  File\|^$' $OUT/test_$i.log | tail -45 | cut -c1-300
done
