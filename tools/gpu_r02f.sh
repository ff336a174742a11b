#!/bin/bash
TAG=${1:-r02f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/ab_tu_variants.py ${2:-144,2192,6288} 32 2>&1 | tail -4 | tee $OUT/ab_tu_variants.txt
for v in $(echo ${2:-144,2192,6288} | tr ',' ' '); do
 for bd in 8 10; do
  OHEVC_TU_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --bit-depth $bd 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'variant': $v, 'bit_depth': $bd, 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['roofline']['kernel_ms'], 'frac': d['roofline']['frac']}))" | tee -a $OUT/bench_ab.jsonl
 done
done
