#!/bin/bash
TAG=${1:-r02s}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_ctx_gpu.py tests/test_stream_gpu.py tests/test_tables_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -6 ) 2>&1 | tee $OUT/pytest_gpu.log
timeout 200 python tools/bench_decode.py --size 1920x1080 --frames 33 2>/dev/null | tail -1 | tee $OUT/bench_decode_1080p_flat.json
