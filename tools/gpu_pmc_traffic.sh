#!/bin/bash
# HBM bytes per launch of the headline kernel: two separate rocprofv3 --pmc passes of the bench command (FETCH_SIZE and WRITE_SIZE do not
# fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), then tools/pmc_traffic.py -> profiles/pmc_traffic.json
TAG=${1:-pmc}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python bench.py --steps 4 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
python tools/pmc_traffic.py $OUT/fetch/p_results.db $OUT/write/p_results.db "tu_idct32_tile1_kernel<unsigned char" | tee $OUT/pmc_traffic.json
find $OUT -name '*.db' -delete
