#!/bin/bash
TAG=${1:-r02w2}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_stream_gpu.py -q -p no:cacheprovider -x -k "pipelined or golden" 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -2 ) 2>&1 | tee $OUT/pytest.log
timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 2>/dev/null | tail -1 > $OUT/flat.json
timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 --natural 2>/dev/null | tail -1 > $OUT/natural.json
timeout 200 python tools/bench_decode.py --size 3840x2160 --frames 17 --bit-depth 10 --natural 2>/dev/null | tail -1 > $OUT/natural4k10.json
timeout 300 python tools/bench_decode.py --size 7680x4320 --frames 9 --bit-depth 10 --natural 2>/dev/null | tail -1 > $OUT/natural8k10.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02w2/*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d.get("bit_exact"), d.get("bit_exact_frame_threads"), {k:(v.get("fps"), v.get("per_picture",{}).get("frame_end_hook_ms")) for k,v in d.items() if isinstance(v,dict)})
PY
