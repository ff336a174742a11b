#!/bin/bash
TAG=${1:-r02t4}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python tools/bench_decode.py --size 1920x1080 --frames 33 --wpp 2>/dev/null | tail -1 > $OUT/decode_1080p_wpp_flat.json
timeout 400 python tools/bench_decode.py --size 1920x1080 --frames 33 --wpp --natural 2>/dev/null | tail -1 > $OUT/decode_1080p_wpp_natural.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02t4/decode_*.json")):
    d=json.load(open(f))
    print(f.split("/")[-1], {k:v for k,v in d.items() if not isinstance(v,dict) and k!="workload"})
    for k,v in d.items():
        if isinstance(v,dict) and "fps" in v and ("slice" in k): print("   %-44s %8.1f fps" % (k, v["fps"]), v.get("per_picture",{}).get("frame_end_hook_ms"))
PY
