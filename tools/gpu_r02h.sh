#!/bin/bash
TAG=${1:-r02h}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_shvc_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -2 ) 2>&1 | tee $OUT/pytest.log
timeout 300 python tools/bench_kernels.py --resident --planes 8 --only shvc 2>/dev/null | grep '^{' > $OUT/bench_shvc.jsonl
python - <<'PY'
import json,sys
for l in open("gpurun_out/%s/bench_shvc.jsonl" % (sys.argv[1] if len(sys.argv)>1 else "r02h")):
    d=json.loads(l); print(d["kernel"][:100], round(d["ms"],4), round(d["frac_hbm_peak"],4))
PY
