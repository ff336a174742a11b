#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench lines and rocprofv3 summaries -> gpurun_out/
# usage (from repo root on the GPU box): bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo" ; rocminfo 2>/dev/null | grep -E 'Marketing Name|gfx950|Compute Unit' | head -6
nproc > $OUT/host_cores.txt
echo "== pytest -m gpu"
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -40 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee $OUT/smoke.log
echo "== plain C host (no python/torch in the process)"
( cd tests/c_host && gcc -O1 host_smoke.c -I../../include -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L../../openhevc_amd -lohevc_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/../../openhevc_amd -Wl,-rpath,/opt/rocm/lib -o host_smoke && timeout 120 ./host_smoke ) 2>&1 | tail -3 | tee $OUT/c_host_smoke.log
echo "== A/B kernel variants"
timeout 600 python tools/ab_tu_variants.py 0,1,16,17 2>&1 | tail -6 | tee $OUT/ab_tu_variants.log
echo "== bench (headline: 32x32 8-bit)"
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -3 | tee $OUT/bench_32x32_8bit.json
echo "== bench variants"
timeout 300 python bench.py --steps 20 --warmup 5 --bit-depth 10 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_32x32_10bit.json
timeout 300 python bench.py --steps 20 --warmup 5 --log2 4 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_16x16_8bit.json
timeout 300 python bench.py --steps 20 --warmup 5 --sparse --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_32x32_8bit_sparse.json
echo "== rocprofv3 kernel trace"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_trace.log 2>&1
python tools/rocpd_summary.py stats $OUT/prof_trace/trace_results.db 2>&1 | cut -c1-150 | tee $OUT/kernel_stats.txt
echo "== rocprofv3 pmc (separate passes)"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_pmc_fetch -o fetch -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_pmc_write -o write -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_write.log 2>&1
python tools/pmc_traffic.py $OUT/prof_pmc_fetch/fetch_results.db $OUT/prof_pmc_write/write_results.db > $OUT/pmc_traffic.json; cat $OUT/pmc_traffic.json
# keep the merge-back small: drop bulky traces, keep CSV summaries
find $OUT -name '*.db' -size +5M -delete 2>/dev/null
du -sh $OUT
