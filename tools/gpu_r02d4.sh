#!/bin/bash
# deblocking from the decoder's maps: the second round of loads (samples, pcm flags, QPs, offsets) issued together
TAG=${1:-r02d4}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_filters_gpu.py tests/test_dbk_maps_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -3 ) 2>&1 | tee $OUT/pytest.log
for i in 1 2; do
  timeout 200 python tools/bench_kernels.py --resident --planes 8 --only deblock 2>/dev/null | grep '^{' >> $OUT/bench_deblock.jsonl
done
python - <<PY
import json
for l in open("$OUT/bench_deblock.jsonl"):
    d=json.loads(l); print(d["kernel"][:70], round(d["ms"],4), round(d["frac_hbm_peak"],4))
PY
