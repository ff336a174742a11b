TAG=${1:-r4s}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_dbk_maps_gpu.py tests/test_filters_gpu.py tests/test_tables_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8 ) 2>&1 | cut -c1-300 | tee $OUT/pytest_subset.log
for v in 0 1; do
  OHEVC_DEBLOCK_VARIANT=$v timeout 600 python tools/kernel_rows.py deblock 2>$OUT/rows_v$v.err > $OUT/rows_deblock_v$v.json
  python - $OUT/rows_deblock_v$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, r in d.items():
    print("variant", sys.argv[2], k, r if isinstance(r, str) else {kk: r[kk] for kk in ("kernel_ms", "achieved", "frac", "checked")})
PY
done
