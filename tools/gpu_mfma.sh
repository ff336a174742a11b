#!/bin/bash
# GPU visit for the matrix-core 32x32 IDCT: layout probe, parity tests, interleaved A/B against the VALU kernel.   bash tools/gpu_mfma.sh <tag> [variants]
TAG=${1:-mfma}; VARS=${2:-144,400,912,1424,1936,400@1024,912@1024}; OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 60 python tools/probe_mfma_layout.py 2>&1 | tail -3 | cut -c1-1500 | tee $OUT/probe.json
timeout 200 python -m pytest tests/test_tu_gpu.py -q -p no:cacheprovider -k "matrix_core" 2>&1 | tail -12 | cut -c1-900 | tee $OUT/pytest.log
timeout 150 python tools/ab_tu_variants.py "$VARS" 32 2>&1 | tail -3 | cut -c1-1500 | tee $OUT/ab.json
