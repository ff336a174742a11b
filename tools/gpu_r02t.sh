#!/bin/bash
TAG=${1:-r02t}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for mode in 2 0; do
  OHHIP_LEVEL_LAUNCH=$mode timeout 200 python tools/bench_decode.py --size 1920x1080 --frames 33 --natural 2>/dev/null | tail -1 | tee $OUT/bench_decode_1080p_natural_mode$mode.json
done
