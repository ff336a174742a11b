#!/bin/bash
TAG=${1:-r02s4}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_filters_gpu.py tests/test_stream_gpu.py tests/test_ctx_gpu.py tests/test_tables_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -3 ) 2>&1 | tee $OUT/pytest.log
for v in 0 2; do
  timeout 300 python tools/bench_kernels.py --resident --planes 8 --only sao --sao-variant $v 2>/dev/null | grep '^{' > $OUT/bench_sao_variant$v.jsonl
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02s4/bench_sao_variant*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(d["kernel"][:100], round(d["ms"],4), round(d["frac_hbm_peak"],4))
PY
CMD="python tools/bench_kernels.py --resident --planes 8 --only sao --sao-variant 0"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  python tools/rocpd_summary.py pmc $OUT/pmc$i/p_results.db sao_wide 2>&1 | cut -c1-220 | tee -a $OUT/counters.txt
done
find $OUT -name '*.db' -delete
