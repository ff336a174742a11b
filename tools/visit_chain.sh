TAG=r4i; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids'
for s in natural flat; do python tools/diag_chain_clocks.py $s 2>/dev/null | grep '^{' | tee -a $OUT/chain_clocks.jsonl; done
timeout 600 python tools/bench_kernels.py --resident --planes 8 --only shvc 2>/dev/null | grep '^{' | tee -a $OUT/bench_kernels_shvc.jsonl | cut -c1-250
( time timeout 1500 python -m pytest tests/test_intra_gpu.py tests/test_shvc_gpu.py tests/test_stream_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v "$NOISE" | tail -15 ) 2>&1 | cut -c1-400 | tee $OUT/pytest_subset.log
for w in 16 32; do
  OHEVC_INTRA_CHAIN_WAVES=$w timeout 900 python bench.py --no-kernels --no-cpu-baseline --no-zscan --steps 20 --decode-hip-only 2> $OUT/bench_w$w.err | tail -1 > $OUT/bench_w$w.json
  python - $OUT/bench_w$w.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.get("decode", {}).get("streams", {}).items():
    print("  decode", sys.argv[1][-8:-5], k, {kk: (vv.get("fps"), vv.get("per_picture", {}).get("frame_end_hook_ms"), vv.get("per_picture", {}).get("launches")) for kk, vv in v.items() if isinstance(vv, dict)})
PY
done
