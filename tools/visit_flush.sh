TAG=${1:-r5k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for kib in 4096 900 600 4096 900 600; do
OHHIP_FLUSH_INTRA_KIB=$kib timeout 900 python bench.py --no-kernels --no-cpu-baseline --no-zscan --steps 20 --decode-hip-only 2> $OUT/bench.err | tail -1 > $OUT/bench_flush_$kib.json
python - $OUT/bench_flush_$kib.json $kib <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.get("decode", {}).get("streams", {}).items():
    print("  flush_kib", sys.argv[2], k, {kk: (vv.get("fps"), vv.get("fps_after_first_pass"), vv.get("per_picture", {}).get("frame_end_hook_ms"), vv.get("per_picture", {}).get("launches")) for kk, vv in v.items() if isinstance(vv, dict)})
PY
done
