#!/bin/bash
TAG=${1:-r02s6}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_filters_gpu.py tests/test_stream_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -2 ) 2>&1 | tee $OUT/pytest.log
timeout 300 python tools/bench_kernels.py --resident --planes 8 --only sao --sao-variant 0 2>/dev/null | grep '^{' > $OUT/bench_sao_variant0.jsonl
python - <<'PY'
import json,glob
for l in open("gpurun_out/r02s6/bench_sao_variant0.jsonl"):
    d=json.loads(l); print(d["kernel"][:100], round(d["ms"],4), round(d["frac_hbm_peak"],4))
PY
