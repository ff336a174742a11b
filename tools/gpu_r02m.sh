#!/bin/bash
TAG=${1:-r02m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -15 ) 2>&1 | tee $OUT/pytest_gpu.log
bash tools/gpu_pmc_traffic.sh $TAG/pmc
cp $OUT/pmc/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python bench.py 2>&1 | tail -1 | tee $OUT/bench.json
timeout 600 python bench.py --bit-depth 10 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_10bit.json
timeout 600 python bench.py --log2 4 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_16x16.json
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_trace.log 2>&1
python tools/rocpd_summary.py stats $OUT/prof_trace/trace_results.db 2>&1 | cut -c1-150 | tee $OUT/kernel_stats.txt
find $OUT -name '*.db' -size +5M -delete 2>/dev/null
