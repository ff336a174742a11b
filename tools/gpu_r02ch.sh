#!/bin/bash
# what one step of the per-level chain costs: kernel durations and stream-idle gaps from a rocprofv3 kernel trace of the flat 1080p stream, 1 thread
TAG=${1:-r02ch}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d /tmp/ch -o t -- python tools/diag_overlap.py decode 1 > $OUT/decode.log 2>&1
tail -1 $OUT/decode.log
python tools/diag_overlap.py analyze /tmp/ch/t_results.db | tee $OUT/overlap.json
python tools/diag_overlap.py chain /tmp/ch/t_results.db | tee $OUT/chain.jsonl
