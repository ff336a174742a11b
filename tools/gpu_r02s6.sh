#!/bin/bash
# SAO wide kernel: rows per lane in flight (variants 4 / 8 / 16 of ohevc_debug_set_sao_variant) next to the shipped form
TAG=${1:-r02s6}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_filters_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -3 ) 2>&1 | tee $OUT/pytest.log
for v in 0 4 8 16 0 4; do
  timeout 200 python tools/bench_kernels.py --resident --planes 8 --only sao --sao-variant $v 2>/dev/null | grep '^{' >> $OUT/bench_sao_variant$v.jsonl
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_sao_variant*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(f.split("variant")[1][:-6], d["kernel"][:90], round(d["ms"],4), round(d["frac_hbm_peak"],4))
PY
