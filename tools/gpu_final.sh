#!/bin/bash
# Round-end GPU check: the whole -m gpu suite, a short fuzz run, kernel benches and the headline bench.   bash tools/gpu_final.sh <tag>
TAG=${1:-final}; OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -8 | cut -c1-800 | tee $OUT/pytest.log
timeout 40 python tools/fuzz_streams.py 15 $RANDOM 2>&1 | grep -v "The cu_qp_delta\|PPS extension\|partially impl" | tail -3 | cut -c1-1500 | tee $OUT/fuzz.log
timeout 60 python tools/bench_kernels.py --planes 8 --only sao 2>&1 | tail -4 | tee $OUT/bench_kernels_sao_x8.jsonl
timeout 60 python tools/bench_kernels.py --only mc 2>&1 | tail -6 | tee $OUT/bench_kernels_mc.jsonl
timeout 120 python bench.py 2>&1 | tail -1 | tee $OUT/bench.json
