#!/bin/bash
# intra neighbour gather with its three loads in flight together: parity on the device + the whole decoder's frame-end hook
TAG=${1:-r02u5}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_stream_gpu.py tests/test_ctx_gpu.py tests/test_tables_gpu.py tests/test_intra_gpu.py tests/test_filters_gpu.py tests/test_dbk_maps_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -3 ) 2>&1 | tee $OUT/pytest.log
timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 2>/dev/null | tail -1 > $OUT/flat.json
timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 --natural 2>/dev/null | tail -1 > $OUT/natural.json
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d.get("bit_exact"), d.get("bit_exact_frame_threads"), {k:(v.get("fps"), v.get("per_picture",{}).get("frame_end_hook_ms"), v.get("per_picture",{}).get("launches")) for k,v in d.items() if isinstance(v,dict) and "hip" in k})
PY
