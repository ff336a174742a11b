TAG=${1:-r4v}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids'
( time timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "$NOISE" | tail -15 ) 2>&1 | cut -c1-400 | tee $OUT/pytest_gpu.log
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/krows -o t -- python $ROOT/tools/kernel_rows.py > $ROOT/$OUT/kernel_rows.json 2> /tmp/krows.log ); tail -2 /tmp/krows.log
python tools/rocpd_summary.py stats /tmp/krows/t_results.db 2>/dev/null | cut -c1-170 | grep "ohevc" | tee $OUT/kernel_rows_rocprof_stats.txt
python - $OUT/kernel_rows.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(k, v if not isinstance(v, dict) else (v["kernel_ms"], v["frac"], v["checked"]))
PY
