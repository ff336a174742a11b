#!/bin/bash
# GPU visit after a kernel change: the whole -m gpu suite, then tools/bench_kernels.py for the kernels named in $2.   bash tools/gpu_kernel_check.sh <tag> <only>
TAG=${1:-kcheck}; OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -8 | cut -c1-900 | tee $OUT/pytest.log
timeout 120 python tools/bench_kernels.py --only "$2" 2>&1 | tail -8 | tee $OUT/bench_kernels.jsonl
