TAG=r4e; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids'
( time timeout 1500 python -m pytest tests/test_intra_gpu.py tests/test_stream_gpu.py tests/test_ctx_gpu.py tests/test_tables_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v "$NOISE" | tail -15 ) 2>&1 | cut -c1-400 | tee $OUT/pytest_subset.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/ch1 -o t -- python $ROOT/tools/diag_overlap.py decode 1 natural > /tmp/ch1.log 2>&1 )
python tools/diag_overlap.py chain /tmp/ch1/t_results.db | tee $OUT/chain_1_natural.jsonl | cut -c1-300
python tools/diag_overlap.py dump /tmp/ch1/t_results.db $OUT/trace_1_natural.csv.gz
for w in 32 16 64; do
  OHEVC_INTRA_CHAIN_WAVES=$w timeout 900 python bench.py --no-kernels --no-cpu-baseline --no-zscan --steps 20 --decode-hip-only 2> $OUT/bench_w$w.err | tail -1 > $OUT/bench_w$w.json
  python - $OUT/bench_w$w.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.get("decode", {}).get("streams", {}).items():
    print("  decode", sys.argv[1][-9:-5], k, {kk: (vv.get("fps"), vv.get("per_picture", {}).get("frame_end_hook_ms"), vv.get("per_picture", {}).get("launches")) for kk, vv in v.items() if isinstance(vv, dict)})
PY
done
for th in 16; do for kind in natural flat; do
  arg=$([ $kind = natural ] && echo natural)
  rm -f /tmp/ft.txt
  OHHIP_TRACE_FRAMES=/tmp/ft.txt python tools/diag_overlap.py decode $th $arg 2>/dev/null | grep fps | tee $OUT/frames_${th}_$kind.jsonl
  python tools/frame_trace.py /tmp/ft.txt | tee -a $OUT/frames_${th}_$kind.jsonl
  gzip -c /tmp/ft.txt > $OUT/frame_trace_${th}_$kind.txt.gz
done; done
