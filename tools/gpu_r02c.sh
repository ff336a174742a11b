#!/bin/bash
# A/B of the residual kernel's epilogue forms: 144 shipped (wave-private 64-sample strips), 640 workgroup-wide 256-sample strips,
# +1024 non-temporal coefficient loads; in-place harness (tools/ab_tu_variants.py) and under bench.py's own conditions (fresh planes)
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/ab_tu_variants.py 144,1168,640,1664 2>&1 | tail -6 | tee $OUT/ab_tu_variants.txt
for v in 144 1168 640 1664; do
  OHEVC_TU_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'variant': $v, 'bit_depth': 8, 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['roofline']['kernel_ms'], 'frac': d['roofline']['frac']}))" | tee -a $OUT/bench_ab.jsonl
done
for v in 144 1664; do
  OHEVC_TU_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --bit-depth 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'variant': $v, 'bit_depth': 10, 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['roofline']['kernel_ms'], 'frac': d['roofline']['frac']}))" | tee -a $OUT/bench_ab.jsonl
  OHEVC_TU_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --log2 4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'variant': $v, 'log2': 4, 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['roofline']['kernel_ms'], 'frac': d['roofline']['frac']}))" | tee -a $OUT/bench_ab.jsonl
done
