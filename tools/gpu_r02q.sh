#!/bin/bash
TAG=${1:-r02q}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 400 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -15 ) 2>&1 | tee $OUT/pytest_gpu.log
timeout 200 python tools/bench_decode.py --size 1920x1080 --frames 33 2>/dev/null | tail -1 | tee $OUT/bench_decode_1080p_flat.json
OHHIP_LEVEL_LAUNCH=0 timeout 200 python tools/bench_decode.py --size 1920x1080 --frames 33 2>/dev/null | tail -1 | tee $OUT/bench_decode_1080p_flat_levels.json
