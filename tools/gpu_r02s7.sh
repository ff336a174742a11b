#!/bin/bash
# SAO wide kernel as a persistent loop (variants 4 = with prefetch of the next job record, 8 = without) next to the shipped form
TAG=${1:-r02s7}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( OHEVC_SAO_GRID=5 timeout 300 python -m pytest tests/test_filters_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -3 ) 2>&1 | tee $OUT/pytest.log
for cfg in "0 2048" "4 2048" "8 2048" "4 4096" "4 1024" "8 4096" "0 2048"; do
  set -- $cfg
  OHEVC_SAO_GRID=$2 timeout 200 python tools/bench_kernels.py --resident --planes 8 --only sao --sao-variant $1 2>/dev/null | grep '^{' >> $OUT/bench_sao_variant$1_grid$2.jsonl
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_sao_variant*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(f.split("variant")[1][:-6], d["kernel"][:50], round(d["ms"],4), round(d["frac_hbm_peak"],4))
PY
