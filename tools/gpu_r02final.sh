#!/bin/bash
# end-of-round check at HEAD: the driver's GPU tier, smoke(), the headline bench + rocprofv3 statistics of the same command
TAG=${1:-r02final}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -6 ) 2>&1 | tee $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 300 python bench.py 2>/dev/null | tail -1 > $OUT/bench.json; cut -c1-400 $OUT/bench.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null 2>&1 ); python tools/rocpd_summary.py stats /tmp/prof_final/t_results.db 2>/dev/null | cut -c1-150 | head -8 | tee $OUT/kernel_stats.txt
