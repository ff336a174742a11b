#!/bin/bash
TAG=${1:-r02v}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export OHHIP_LEVEL_LAUNCH=3
for kind in natural flat; do
  arg=""; [ $kind = natural ] && arg=natural
  timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/$kind -o t -- python tools/diag_overlap.py decode 1 $arg > $OUT/$kind.log 2>&1
  tail -1 $OUT/$kind.log | tee -a $OUT/summary.txt
  python tools/rocpd_summary.py stats $OUT/$kind/t_results.db | cut -c1-140 | tee -a $OUT/summary.txt
done
find $OUT -name '*.db' -delete
