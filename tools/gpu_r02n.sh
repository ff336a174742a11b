#!/bin/bash
TAG=${1:-r02n}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for th in 1 16; do
  python tools/diag_overlap.py decode $th 2>/dev/null | tail -1 | tee -a $OUT/overlap.jsonl
  timeout 600 rocprofv3 --kernel-trace -d $OUT/t$th -o t -- python tools/diag_overlap.py decode $th > $OUT/t$th.log 2>&1
  tail -1 $OUT/t$th.log | tee -a $OUT/overlap.jsonl
  python tools/diag_overlap.py analyze $OUT/t$th/t_results.db | tee -a $OUT/overlap.jsonl
done
OHHIP_LEVEL_LAUNCH=1 python tools/diag_overlap.py decode 16 2>/dev/null | tail -1 | tee -a $OUT/overlap.jsonl
find $OUT -name '*.db' -delete
