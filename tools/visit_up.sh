TAG=${1:-r5c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for kib in 3072 0 3072 0; do
OHEVC_PREWARM_KIB=$kib timeout 900 python bench.py --no-kernels --no-cpu-baseline --no-zscan --steps 20 --decode-hip-only 2> $OUT/bench.err | tail -1 > $OUT/bench_prewarm_$kib.json
python - $OUT/bench_prewarm_$kib.json $kib <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.get("decode", {}).get("streams", {}).items():
    print("  prewarm", sys.argv[2], k, {kk: (vv.get("fps"), vv.get("fps_after_first_pass"), vv.get("per_picture", {}).get("frame_end_hook_ms")) for kk, vv in v.items() if isinstance(vv, dict)})
PY
done
