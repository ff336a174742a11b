TAG=r4o; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
timeout 900 python bench.py --no-kernels --no-cpu-baseline --no-zscan --steps 20 --decode-hip-only 2> $OUT/bench.err | tail -1 > $OUT/bench_decode_defaults.json
python - $OUT/bench_decode_defaults.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.get("decode", {}).get("streams", {}).items():
    print("  decode", k, {kk: (vv.get("fps"), vv.get("per_picture", {}).get("frame_end_hook_ms"), vv.get("per_picture", {}).get("launches")) for kk, vv in v.items() if isinstance(vv, dict)})
PY
bash tools/gpu.sh $TAG -- bench_prof --no-kernels -- pmc
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/krows -o t -- python $ROOT/tools/kernel_rows.py > $ROOT/$OUT/kernel_rows.json 2> /tmp/krows.log )
python tools/rocpd_summary.py stats /tmp/krows/t_results.db 2>/dev/null | cut -c1-170 | head -40 | tee $OUT/kernel_rows_rocprof_stats.txt
tail -c 3000 $OUT/kernel_rows.json | head -c 600
