#!/bin/bash
# after the chroma second-segment load fix (ASAN finding): the filter / stream suites on the device once more
TAG=${1:-r02end2}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_filters_gpu.py tests/test_dbk_maps_gpu.py tests/test_stream_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -3 ) 2>&1 | tee $OUT/pytest.log
timeout 100 python tools/bench_kernels.py --resident --planes 8 --only deblock 2>/dev/null | grep '^{' > $OUT/bench_deblock.jsonl
python - <<PY
import json
for l in open("$OUT/bench_deblock.jsonl"):
    d=json.loads(l); print(d["kernel"][:70], round(d["ms"],4), round(d["frac_hbm_peak"],4))
PY
