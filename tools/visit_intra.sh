TAG=${1:-r4t}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids'
( time timeout 900 python -m pytest tests/test_intra_gpu.py tests/test_ctx_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v "$NOISE" | tail -6 ) 2>&1 | cut -c1-300 | tee $OUT/pytest_subset.log
for s in natural flat; do python tools/diag_chain_clocks.py $s 2>/dev/null | grep '^{' | tee -a $OUT/chain_clocks.jsonl | cut -c1-600; done
timeout 600 python tools/kernel_rows.py intra 2>$OUT/rows.err > $OUT/rows_intra.json
python - $OUT/rows_intra.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, r in d.items():
    print(k, r if isinstance(r, str) else {kk: r[kk] for kk in ("kernel_ms", "achieved", "frac", "checked")})
PY
timeout 900 python bench.py --no-kernels --no-cpu-baseline --no-zscan --steps 20 --decode-hip-only 2> $OUT/bench.err | tail -1 > $OUT/bench.json
python - $OUT/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("headline", d.get("value"), d.get("roofline", {}).get("frac"))
for k, v in d.get("decode", {}).get("streams", {}).items():
    print("  decode", k, {kk: (vv.get("fps"), vv.get("per_picture", {}).get("frame_end_hook_ms"), vv.get("per_picture", {}).get("launches")) for kk, vv in v.items() if isinstance(vv, dict)})
PY
