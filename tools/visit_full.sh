TAG=${1:-full}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids'
( time timeout 1800 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "$NOISE" > $OUT/pytest_gpu_complete.log; tail -15 $OUT/pytest_gpu_complete.log ) 2>&1 | cut -c1-400 | tee $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$NOISE" | tail -2 | tee $OUT/smoke.log
timeout 1500 python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json; cut -c1-300 $OUT/bench.json; tail -3 $OUT/bench.err | grep -v "$NOISE"
nproc > $OUT/host_cores.txt
