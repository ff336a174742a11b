#!/bin/bash
# fused prediction + residual per level: parity (suites + fuzz incl. frame threads) and A/B on the whole decoder
TAG=${1:-r02u4}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_stream_gpu.py tests/test_ctx_gpu.py tests/test_tables_gpu.py tests/test_intra_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -3 ) 2>&1 | tee $OUT/pytest.log
( OHHIP_LEVEL_LAUNCH=0 timeout 200 python tools/fuzz_streams.py 120 4711 2>&1 | tail -1 ) | tee $OUT/fuzz_levels_fused.json | cut -c1-300
for f in 1 0; do
  OHEVC_FUSE_INTRA=$f timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 2>/dev/null | tail -1 > $OUT/flat_fuse$f.json
  OHEVC_FUSE_INTRA=$f timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 --natural 2>/dev/null | tail -1 > $OUT/natural_fuse$f.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02u4/*_fuse*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d.get("bit_exact"), d.get("bit_exact_frame_threads"), {k:(v.get("fps"), v.get("per_picture",{}).get("frame_end_hook_ms"), v.get("per_picture",{}).get("launches")) for k,v in d.items() if isinstance(v,dict) and "hip" in k})
PY
