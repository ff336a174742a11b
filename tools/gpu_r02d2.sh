#!/bin/bash
TAG=${1:-r02d3}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_dbk_maps_gpu.py tests/test_filters_gpu.py tests/test_stream_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -2 ) 2>&1 | tee $OUT/pytest.log
timeout 300 python tools/bench_kernels.py --resident --planes 8 --only deblock 2>/dev/null | grep '^{' > $OUT/bench_deblock.jsonl
python - <<'PY'
import json,sys
for l in open("gpurun_out/%s/bench_deblock.jsonl" % sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r02d3/bench_deblock.jsonl"):
    d=json.loads(l); print(d["kernel"][:100], round(d["ms"],4), round(d["frac_hbm_peak"],4))
PY
