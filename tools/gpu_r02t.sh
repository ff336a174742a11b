#!/bin/bash
# thread modes of the whole decoder: 1 thread, frame threads, slice (WPP) threads, both - reference C vs HIP back end
TAG=${1:-r02t3}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python tools/bench_decode.py --size 1920x1080 --frames 33 --wpp 2>/dev/null | tail -1 > $OUT/decode_1080p_wpp_flat.json
timeout 400 python tools/bench_decode.py --size 1920x1080 --frames 33 --wpp --natural 2>/dev/null | tail -1 > $OUT/decode_1080p_wpp_natural.json
timeout 500 python tools/bench_decode.py --size 3840x2160 --frames 17 --bit-depth 10 --wpp 2>/dev/null | tail -1 > $OUT/decode_4k10_wpp_flat.json
timeout 500 python tools/bench_decode.py --size 3840x2160 --frames 17 --bit-depth 10 --wpp --natural 2>/dev/null | tail -1 > $OUT/decode_4k10_wpp_natural.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02t3/decode_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], "bit_exact", d.get("bit_exact"), d.get("bit_exact_frame_threads"), d.get("bit_exact_slice_threads"))
    for k,v in d.items():
        if isinstance(v,dict) and "fps" in v: print("   %-44s %8.1f fps" % (k, v["fps"]), v.get("per_picture",{}).get("frame_end_hook_ms"), v.get("per_picture",{}).get("launches"))
PY
