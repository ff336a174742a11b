#!/bin/bash
TAG=${1:-r02u}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_ctx_gpu.py tests/test_stream_gpu.py tests/test_tables_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -6 ) 2>&1 | tee $OUT/pytest_gpu.log
for mode in 2 3 0; do
  OHHIP_LEVEL_LAUNCH=$mode timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 2>/dev/null | tail -1 > $OUT/flat_mode$mode.json
  OHHIP_LEVEL_LAUNCH=$mode timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 --natural 2>/dev/null | tail -1 > $OUT/natural_mode$mode.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out//*_mode*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], {k:(v.get("fps"), v.get("per_picture",{}).get("frame_end_hook_ms"), v.get("per_picture",{}).get("launches")) for k,v in d.items() if isinstance(v,dict) and "hip" in k})
PY
