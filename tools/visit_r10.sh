# round 4: rocprofv3 --kernel-trace --stats of the bench command at the final code (the headline kernel's average duration beside the line's own number)
TAG=${1:-r10}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/benchprof -o t -- python $ROOT/bench.py --no-decode > $ROOT/$OUT/bench_under_rocprof.json 2> /tmp/benchprof.log ); tail -1 /tmp/benchprof.log | cut -c1-160
python tools/rocpd_summary.py stats /tmp/benchprof/t_results.db 2>/dev/null | cut -c1-170 | head -30 | tee $OUT/kernel_stats.txt
python -c "
import json; d=json.loads(open('$OUT/bench_under_rocprof.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel_ms'], d['roofline']['frac'])"
