#!/bin/bash
# SAO wide kernel: neighbour loads issued together (no loads inside the picture-edge branches), row loop without the rules call for blocks
# no position rule can touch
TAG=${1:-r02s9}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_filters_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -3 ) 2>&1 | tee $OUT/pytest.log
for v in 0 0; do
  timeout 200 python tools/bench_kernels.py --resident --planes 8 --only sao 2>/dev/null | grep '^{' >> $OUT/bench_sao.jsonl
done
python - <<PY
import json,glob
for l in open("$OUT/bench_sao.jsonl"):
    d=json.loads(l); print(d["kernel"][:50], round(d["ms"],4), round(d["frac_hbm_peak"],4))
PY
