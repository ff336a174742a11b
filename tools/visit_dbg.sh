TAG=${1:-r5j}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids'
for i in 1 2 3 4; do
timeout 1800 python -X faulthandler -m pytest tests/test_ctx_gpu.py tests/test_dist_gpu.py tests/test_stream_gpu.py tests/test_tables_gpu.py -m gpu -q -p no:cacheprovider -x -v 2>&1 | grep -v "$NOISE" > $OUT/pytest_gpu_full_$i.log
echo "run $i: $(grep -c PASSED $OUT/pytest_gpu_full_$i.log) passed; $(grep -c 'Fatal\|dumped\|Aborted\|Segmentation' $OUT/pytest_gpu_full_$i.log) crash lines"
done
