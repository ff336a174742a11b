TAG=${1:-r4l}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift; timeout 1500 python tools/bench_decode.py "$@" 2>/dev/null | tail -1 > $OUT/decode_$name.json
  python - $OUT/decode_$name.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], d.get("workload"), "bit_exact", d.get("bit_exact"), d.get("bit_exact_frame_threads"), "sse==c", d.get("reference_sse_equals_reference_c"))
for k, v in d.items():
    if isinstance(v, dict) and "fps" in v:
        pp = v.get("per_picture", {})
        print("   ", k, v["fps"], "fps", pp.get("frame_end_hook_ms", ""), pp.get("launches", ""), pp.get("upload_kib", ""))
PY
}
run 1080p_natural_wpp --natural --wpp --frames 33 --cpu-threads 8 --passes 4
run 1080p_flat_wpp --wpp --frames 33 --cpu-threads 8 --passes 4
run 4k_main10_natural --size 3840x2160 --bit-depth 10 --natural --frames 9 --cpu-threads 8 --passes 3
run 4k_main10_qp22 --size 3840x2160 --bit-depth 10 --qp22 --frames 9 --cpu-threads 8 --passes 2
run 8k_main10_natural --size 7680x4320 --bit-depth 10 --natural --frames 5 --cpu-threads 8 --passes 2
run 8k_main10_qp22 --size 7680x4320 --bit-depth 10 --qp22 --frames 3 --cpu-threads 8 --passes 2
