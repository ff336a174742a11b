#!/bin/bash
TAG=${1:-r02dist}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_dist_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -12 ) 2>&1 | tee $OUT/pytest.log
