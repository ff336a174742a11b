#!/bin/bash
# a longer fuzz of the shipped configuration (executor chosen per picture, device-derived deblocking, mc4, wide SAO, fused levels)
TAG=${1:-r02fz}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 260 python tools/fuzz_streams.py 240 90210 2>&1 | tail -1 ) | tee $OUT/fuzz_default.json | cut -c1-300
