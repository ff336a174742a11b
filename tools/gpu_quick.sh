#!/bin/bash
# Short GPU visit: parity tests + kernel A/B only.   bash tools/gpu_quick.sh <tag> [ab variants]
TAG=${1:-quick}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -40 | tee $OUT/pytest_gpu.log
echo "== A/B kernel variants"
timeout 900 python tools/ab_tu_variants.py $2 2>&1 | tail -8 | tee $OUT/ab_tu_variants.log
