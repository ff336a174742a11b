TAG=r4q; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
for m in 2 0; do
  OHHIP_LEVEL_LAUNCH=$m timeout 900 python bench.py --no-kernels --no-cpu-baseline --no-zscan --steps 20 --decode-hip-only 2> $OUT/bench.err | tail -1 > $OUT/bench_levelmode_$m.json
  python - $OUT/bench_levelmode_$m.json $m <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.get("decode", {}).get("streams", {}).items():
    print("  decode level_launch", sys.argv[2], k, {kk: (vv.get("fps"), vv.get("per_picture", {}).get("frame_end_hook_ms"), vv.get("per_picture", {}).get("launches")) for kk, vv in v.items() if isinstance(vv, dict)})
PY
done
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/krows -o t -- python $ROOT/tools/kernel_rows.py > $ROOT/$OUT/kernel_rows.json 2> /tmp/krows.log ); tail -2 /tmp/krows.log
python tools/rocpd_summary.py stats /tmp/krows/t_results.db 2>/dev/null | cut -c1-170 | grep "ohevc" | tee $OUT/kernel_rows_rocprof_stats.txt
python - $OUT/kernel_rows.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(k, v if not isinstance(v, dict) else (v["kernel_ms"], v["frac"], v["checked"]))
PY
