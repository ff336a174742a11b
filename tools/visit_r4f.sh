TAG=r4f; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids'
( time timeout 1500 python -m pytest tests/test_intra_gpu.py tests/test_shvc_gpu.py tests/test_stream_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v "$NOISE" | tail -15 ) 2>&1 | cut -c1-400 | tee $OUT/pytest_subset.log
for v in 0 1; do OHEVC_UPSAMPLE_VARIANT=$v timeout 600 python tools/bench_kernels.py --resident --planes 8 --only shvc 2>/dev/null | grep '^{' | tee -a $OUT/bench_kernels_shvc_variant$v.jsonl | cut -c1-250; done
for acq in 0 1; do for w in 16 32 64; do
  OHEVC_CHAIN_AGENT_ACQUIRE=$acq OHEVC_INTRA_CHAIN_WAVES=$w timeout 900 python bench.py --no-kernels --no-cpu-baseline --no-zscan --steps 20 --decode-hip-only 2> $OUT/bench_a${acq}_w$w.err | tail -1 > $OUT/bench_a${acq}_w$w.json
  python - $OUT/bench_a${acq}_w$w.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.get("decode", {}).get("streams", {}).items():
    print("  decode", sys.argv[1][-11:-5], k, {kk: (vv.get("fps"), vv.get("per_picture", {}).get("frame_end_hook_ms"), vv.get("per_picture", {}).get("launches")) for kk, vv in v.items() if isinstance(vv, dict)})
PY
done; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/ch1 -o t -- python $ROOT/tools/diag_overlap.py decode 1 natural > /tmp/ch1.log 2>&1 )
python tools/diag_overlap.py chain /tmp/ch1/t_results.db | tee $OUT/chain_1_natural.jsonl | cut -c1-300
python tools/diag_overlap.py dump /tmp/ch1/t_results.db $OUT/trace_1_natural.csv.gz
