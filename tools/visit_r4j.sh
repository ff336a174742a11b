TAG=r4j; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for s in natural flat; do python tools/diag_chain_clocks.py $s 2>/dev/null | grep '^{' | tee -a $OUT/chain_clocks.jsonl; done
timeout 600 python tools/bench_kernels.py --resident --planes 8 --only shvc 2>/dev/null | grep '^{' | tee -a $OUT/bench_kernels_shvc.jsonl | cut -c1-250
timeout 600 python -m pytest tests/test_shvc_gpu.py tests/test_intra_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
