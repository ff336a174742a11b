TAG=${1:-r5b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
( cd /tmp && OHHIP_TRACE_FRAMES=/tmp/ft.txt timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -d /tmp/ov16h -o t -- python $ROOT/tools/diag_overlap.py decode 16 natural > /tmp/ov16h.log 2>&1 )
grep fps /tmp/ov16h.log | tee $OUT/overlap_16_natural_hip.jsonl
python tools/diag_overlap.py dump /tmp/ov16h/t_results.db $OUT/trace_16_natural_hip.csv.gz
python tools/frame_trace.py /tmp/ft.txt --dump 2>/dev/null > $OUT/frame_trace_16_natural_dump.txt
cp /tmp/ft.txt $OUT/frame_trace_raw.txt
ls -la $OUT
