# round 4, closing visit: the complete device suite, smoke, the full bench line, a short fuzz campaign of each kind on the device
TAG=${1:-r8}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids\|IRAP'
( time timeout 900 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "$NOISE" > $OUT/pytest_gpu_complete.log; tail -8 $OUT/pytest_gpu_complete.log ) 2>&1 | cut -c1-400 | tee $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$NOISE" | tail -2 | tee $OUT/smoke.log
timeout 900 python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json; cut -c1-300 $OUT/bench.json; tail -3 $OUT/bench.err | grep -v "$NOISE"
FUZZ_BACKEND=hip timeout 100 python tools/fuzz_streams.py 35 808 2> /dev/null | tail -2 | cut -c1-600 | tee $OUT/fuzz_streams_device.txt
FUZZ_BACKEND=hip timeout 100 python tools/fuzz_shvc.py 25 909 2> /dev/null | tail -2 | cut -c1-600 | tee $OUT/fuzz_shvc_device.txt
nproc > $OUT/host_cores.txt
