TAG=${1:-r4u}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_shvc_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4 ) 2>&1 | cut -c1-300 | tee $OUT/pytest_subset.log
timeout 600 python tools/bench_kernels.py --resident --planes 8 --only shvc 2>/dev/null | grep '^{' | tee $OUT/bench_kernels_shvc.jsonl | cut -c1-260
