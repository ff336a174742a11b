#!/bin/bash
# Measurement visit: per-kernel bench + whole-decoder runs incl. the reference's slice threads (WPP).   bash tools/gpu_measure.sh <tag>
TAG=${1:-measure}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host_cores.txt
F='The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids'
timeout 400 python tools/bench_kernels.py 2>&1 | grep -v "$F" > $OUT/per_kernel_4k.jsonl; tail -3 $OUT/per_kernel_4k.jsonl | cut -c1-300
timeout 500 python tools/bench_decode.py --size 1920x1080 --frames 17 --wpp --cpu-threads 8 2>&1 | grep -v "$F" | tail -1 > $OUT/decode_1080p_wpp.json; cut -c1-400 $OUT/decode_1080p_wpp.json
timeout 700 python tools/bench_decode.py --size 3840x2160 --frames 9 --bit-depth 10 --wpp --cpu-threads 8 2>&1 | grep -v "$F" | tail -1 > $OUT/decode_4k_main10_wpp.json; cut -c1-400 $OUT/decode_4k_main10_wpp.json
timeout 400 python tools/bench_decode.py --size 1920x1080 --frames 9 --chroma-format 3 --cpu-threads 8 2>&1 | grep -v "$F" | tail -1 > $OUT/decode_1080p_444.json; cut -c1-300 $OUT/decode_1080p_444.json
