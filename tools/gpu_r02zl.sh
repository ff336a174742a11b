#!/bin/bash
TAG=${1:-r02zl}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_mc_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -3 ) 2>&1 | tee $OUT/pytest_mc.log
timeout 300 python tools/bench_kernels.py --resident --planes 8 --only mc --mc-variant 4 2>/dev/null | grep '^{' > $OUT/bench_mc_variant4.jsonl
python - <<'PY'
import json,glob,sys
for f in sorted(glob.glob("gpurun_out/"+sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r02zl/bench_mc_variant4.jsonl")):
    for l in open(f):
        d=json.loads(l); print(d["kernel"][:90], round(d["ms"],4), round(d["frac_hbm_peak"],4))
PY
( timeout 300 python -m pytest tests/test_stream_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -2 ) 2>&1 | tee gpurun_out/r02zl/pytest_streams.log
