#!/bin/bash
# per-kernel HBM numbers: tools/bench_kernels.py --resident --planes 8 (rings > 2 GiB), its rocprofv3 kernel statistics and the two PMC passes
TAG=${1:-r02x}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python tools/bench_kernels.py --resident --planes 8"
timeout 400 $CMD 2>&1 | grep '^{' | tee $OUT/bench_kernels_resident_x8.jsonl
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
python tools/rocpd_summary.py stats $OUT/trace/t_results.db | cut -c1-150 | tee $OUT/kernel_stats.txt
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
python tools/pmc_per_kernel.py $OUT/trace/t_results.db $OUT/fetch/p_results.db $OUT/write/p_results.db | tee $OUT/pmc_per_config.jsonl
find $OUT -name '*.db' -delete
