#!/bin/bash
# round 2, visit z: the matrix-core motion compensation (mc4) - parity on the device, A/B against mc3 out of HBM
TAG=${1:-r02z}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_mc_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -6 ) 2>&1 | tee $OUT/pytest_mc.log
for v in 3 4; do
  timeout 300 python tools/bench_kernels.py --resident --planes 8 --only mc --mc-variant $v 2>/dev/null | grep '^{' > $OUT/bench_mc_variant$v.jsonl
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/*/bench_mc_variant*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(d["kernel"][:90], round(d["ms"],4), round(d["frac_hbm_peak"],4))
PY
