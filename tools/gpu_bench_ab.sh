#!/bin/bash
# bench.py under its own conditions (fresh random prediction planes) with the shipped residual kernel and with alternatives.   bash tools/gpu_bench_ab.sh <tag> "<variants>"
TAG=${1:-benchab}; OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for v in "" $2; do
  echo "variant=${v:-shipped}"
  OHEVC_TU_VARIANT=$v timeout 120 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({k: d[k] for k in ('value','ms_per_step','roofline')}))" | tee -a $OUT/bench_ab.jsonl
done
