#!/bin/bash
# SAO wide form: parity on the device + A/B against the LDS-window form out of HBM
TAG=${1:-r02s3}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_filters_gpu.py tests/test_stream_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -3 ) 2>&1 | tee $OUT/pytest.log
for v in 0 2; do
  timeout 300 python tools/bench_kernels.py --resident --planes 8 --only sao --sao-variant $v 2>/dev/null | grep '^{' > $OUT/bench_sao_variant$v.jsonl
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02s*/bench_sao_variant*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(d["kernel"][:100], round(d["ms"],4), round(d["frac_hbm_peak"],4))
PY
