#!/bin/bash
TAG=${1:-r02v2}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for d in 0 1; do
  if [ $d = 1 ]; then export OHHIP_DEFER_DOWNLOAD=1; else unset OHHIP_DEFER_DOWNLOAD; fi
  timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 2>/dev/null | tail -1 > $OUT/flat_defer$d.json
  timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 --natural 2>/dev/null | tail -1 > $OUT/natural_defer$d.json
  timeout 200 python tools/bench_decode.py --size 3840x2160 --frames 17 --bit-depth 10 --natural 2>/dev/null | tail -1 > $OUT/natural4k10_defer$d.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02v2/*_defer*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d.get("bit_exact"), d.get("bit_exact_frame_threads"), {k:(v.get("fps"), v.get("per_picture",{}).get("frame_end_hook_ms")) for k,v in d.items() if isinstance(v,dict) and "hip" in k})
PY
