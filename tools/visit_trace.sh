# one visit: device + HIP-API timelines of the hooked decoder with 16 frame threads (tools/diag_overlap.py dump -> gpurun_out/<tag>/*.csv.gz)
TAG=${1:-trace}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
for kind in natural flat; do
  arg=$([ $kind = natural ] && echo natural)
  python tools/diag_overlap.py decode 16 $arg 2>/dev/null | grep fps | tee $OUT/plain_16_$kind.jsonl
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/ov16$kind -o t -- python $ROOT/tools/diag_overlap.py decode 16 $arg > /tmp/ov16$kind.log 2>&1 )
  grep fps /tmp/ov16$kind.log | tee $OUT/overlap_16_$kind.jsonl
  python tools/diag_overlap.py analyze /tmp/ov16$kind/t_results.db | tee -a $OUT/overlap_16_$kind.jsonl
  python tools/diag_overlap.py dump /tmp/ov16$kind/t_results.db $OUT/trace_16_$kind.csv.gz
done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -d /tmp/ov16h -o t -- python $ROOT/tools/diag_overlap.py decode 16 natural > /tmp/ov16h.log 2>&1 )
grep fps /tmp/ov16h.log | tee $OUT/overlap_16_natural_hip.jsonl
python tools/diag_overlap.py dump /tmp/ov16h/t_results.db $OUT/trace_16_natural_hip.csv.gz
ls -la $OUT
