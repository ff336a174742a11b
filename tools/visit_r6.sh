# round 4, last visit: the SHVC stream tests first (new), then the complete device suite, smoke, the bench line (with the shvc row)
TAG=${1:-r6}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids\|IRAP'
( time timeout 400 python -X faulthandler -m pytest tests/test_shvc_stream_gpu.py -q -p no:cacheprovider 2>&1 | grep -v "$NOISE" > $OUT/pytest_shvc_complete.log; tail -8 $OUT/pytest_shvc_complete.log ) 2>&1 | cut -c1-400 | tee $OUT/pytest_shvc.log
timeout 200 python tools/bench_shvc.py --size 1920x1088 --frames 17 --passes 3 2> /dev/null | tail -1 > $OUT/bench_shvc_1080p.json; cut -c1-700 $OUT/bench_shvc_1080p.json
( time timeout 900 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_shvc_stream_gpu.py 2>&1 | grep -v "$NOISE" > $OUT/pytest_gpu_complete.log; tail -8 $OUT/pytest_gpu_complete.log ) 2>&1 | cut -c1-400 | tee $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$NOISE" | tail -2 | tee $OUT/smoke.log
timeout 900 python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json; cut -c1-300 $OUT/bench.json; tail -3 $OUT/bench.err | grep -v "$NOISE"
timeout 200 python tools/bench_shvc.py --size 3840x2176 --frames 9 --passes 2 2> /dev/null | tail -1 > $OUT/bench_shvc_4k.json; cut -c1-700 $OUT/bench_shvc_4k.json
nproc > $OUT/host_cores.txt
