#!/bin/bash
# executor of the intra chain under frame threads: one launch per level (default) vs the persistent level kernel (one launch for the chain)
TAG=${1:-r02lv}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for m in 2 1; do
  OHHIP_LEVEL_LAUNCH=$m timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 2>/dev/null | tail -1 > $OUT/flat_mode$m.json
  OHHIP_LEVEL_LAUNCH=$m timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 --natural 2>/dev/null | tail -1 > $OUT/natural_mode$m.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d.get("bit_exact"), d.get("bit_exact_frame_threads"), {k:(v.get("fps"), v.get("per_picture",{}).get("frame_end_hook_ms"), v.get("per_picture",{}).get("launches")) for k,v in d.items() if isinstance(v,dict) and "hip" in k})
PY
