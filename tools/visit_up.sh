TAG=${1:-r5c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids'
timeout 900 python bench.py --no-kernels --no-cpu-baseline --no-zscan --steps 20 2> $OUT/bench.err | tail -1 > $OUT/bench.json
python - $OUT/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("headline", d.get("value"), d.get("roofline", {}).get("frac"))
for k, v in d.get("decode", {}).get("streams", {}).items():
    print("  decode", k, {kk: (vv.get("fps"), vv.get("fps_after_first_pass"), vv.get("per_picture", {}).get("frame_end_hook_ms")) for kk, vv in v.items() if isinstance(vv, dict)})
PY
OHHIP_TRACE_FRAMES=/tmp/ft.txt python tools/diag_overlap.py decode 16 natural 2>/dev/null | grep fps
python tools/frame_trace.py /tmp/ft.txt --dump 2>/dev/null > $OUT/frame_trace_16_natural_dump.txt; head -1 $OUT/frame_trace_16_natural_dump.txt | cut -c1-400
