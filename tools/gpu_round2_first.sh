#!/bin/bash
# First GPU-box visit of the next round: what round 1 left unmeasured on the device.   bash tools/gpu_round2_first.sh [tag]
#  1. the full -m gpu suite (it now includes the SAO interior / ring variant, so far only run over the CPU emulator)
#  2. SAO A/B, interleaved twice, single plane and eight planes per launch -> decide the default (include/ohevc_debug.h)
#  3. the headline bench line + rocprofv3 kernel statistics of the same command
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
for rep in 1 2; do
  for v in 0 1; do
    timeout 300 python tools/bench_kernels.py --only sao --sao-variant $v 2>&1 | grep '^{' | tee -a $OUT/sao_ab_x1.jsonl
    timeout 300 python tools/bench_kernels.py --only sao --sao-variant $v --planes 8 2>&1 | grep '^{' | tee -a $OUT/sao_ab_x8.jsonl
  done
done
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee $OUT/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_trace.log 2>&1
python tools/rocpd_summary.py stats $OUT/prof_trace/trace_results.db 2>&1 | cut -c1-150 | tee $OUT/kernel_stats.txt
find $OUT -name '*.db' -size +5M -delete 2>/dev/null
