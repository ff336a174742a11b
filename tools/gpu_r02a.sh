#!/bin/bash
# round 2, first device visit: full -m gpu suite (BASELINE geometry + MD5 SEI tests are new), headline bench + rocprofv3, streaming
# ceilings of the box (tools/hbm_probe), SAO interior/ring A/B
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -25 ) 2>&1 | tee $OUT/pytest_gpu.log
timeout 300 tools/hbm_probe 2 2>&1 | tee $OUT/hbm_probe.jsonl
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee $OUT/bench.json
for rep in 1 2; do
  for v in 0 1; do
    timeout 300 python tools/bench_kernels.py --only sao --sao-variant $v --planes 8 2>&1 | grep '^{' | tee -a $OUT/sao_ab_x8.jsonl
  done
done
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_trace.log 2>&1
python tools/rocpd_summary.py stats $OUT/prof_trace/trace_results.db 2>&1 | cut -c1-150 | tee $OUT/kernel_stats.txt
find $OUT -name '*.db' -size +5M -delete 2>/dev/null
