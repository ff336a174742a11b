# round 4: SHVC on the device again (ratio 1 + phase alignment fixed), the dense-at-size config tests, kernel shares of a two-layer decode, two-layer fuzzing on the device
TAG=${1:-r7}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids\|IRAP'
( time timeout 500 python -X faulthandler -m pytest tests/test_shvc_stream_gpu.py tests/test_stream_gpu.py -k "shvc or dense_residual" -q -p no:cacheprovider 2>&1 | grep -v "$NOISE" > $OUT/pytest_shvc_and_dense_complete.log; tail -6 $OUT/pytest_shvc_and_dense_complete.log ) 2>&1 | cut -c1-400 | tee $OUT/pytest_shvc_and_dense.log
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/shvcprof -o t -- python $ROOT/tools/bench_shvc.py --size 1920x1088 --frames 17 --passes 2 --kinds hip > $ROOT/$OUT/bench_shvc_under_rocprof.json 2> /tmp/shvcprof.log ); tail -1 /tmp/shvcprof.log | cut -c1-200
python tools/rocpd_summary.py stats /tmp/shvcprof/t_results.db 2>/dev/null | cut -c1-170 | head -40 | tee $OUT/shvc_decode_kernel_stats.txt
FUZZ_BACKEND=hip timeout 120 python tools/fuzz_shvc.py 50 77 2> /dev/null | tail -4 | cut -c1-1500 | tee $OUT/fuzz_shvc_device.txt
