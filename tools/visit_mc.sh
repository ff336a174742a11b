TAG=${1:-r4x}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_mc_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4 ) 2>&1 | cut -c1-300 | tee $OUT/pytest_subset.log
timeout 600 python tools/kernel_rows.py mc 2>$OUT/rows.err > $OUT/rows_mc.json
python - $OUT/rows_mc.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, r in d.items():
    print(k, r if isinstance(r, str) else {kk: r[kk] for kk in ("kernel_ms", "achieved", "frac", "checked")})
PY
