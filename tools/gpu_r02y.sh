#!/bin/bash
# round 2, visit y: full GPU suite (14 bit, device-derived deblocking, picture export / import), the f3 A/B on the whole decoder,
# bench.py both modes.  Every command carries its own timeout.
TAG=${1:-r02y}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -8 ) 2>&1 | tee $OUT/pytest_gpu.log
timeout 200 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench.json; cut -c1-600 $OUT/bench.json
for f in 1 0; do
  OHHIP_DEVICE_FILTERS=$f timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 2>/dev/null | tail -1 > $OUT/flat_devfilters$f.json
  OHHIP_DEVICE_FILTERS=$f timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 --natural 2>/dev/null | tail -1 > $OUT/natural_devfilters$f.json
done
OHHIP_DEVICE_FILTERS=1 timeout 200 python tools/bench_decode.py --size 3840x2160 --frames 17 --bit-depth 10 --natural 2>/dev/null | tail -1 > $OUT/natural_4k10_devfilters1.json
OHHIP_DEVICE_FILTERS=0 timeout 200 python tools/bench_decode.py --size 3840x2160 --frames 17 --bit-depth 10 --natural 2>/dev/null | tail -1 > $OUT/natural_4k10_devfilters0.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/*/*devfilters*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d.get("bit_exact"), {k:(v.get("fps"), v.get("per_picture",{}).get("frame_end_hook_ms"), v.get("per_picture",{}).get("launches"), v.get("per_picture",{}).get("upload_bytes")) for k,v in d.items() if isinstance(v,dict) and "hip" in k})
PY
timeout 300 python bench.py --mode frames --steps 2 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_frames_1gpu.json | cut -c1-700
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_y -o y -- python $GRAFT_REPO_ROOT/tools/bench_decode.py --size 1920x1080 --frames 17 --natural > /dev/null 2>&1 ); f=$(find /tmp/prof_y -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -25 "$f" > $OUT/kernel_stats_decode_1080p_natural.csv && head -12 $OUT/kernel_stats_decode_1080p_natural.csv | cut -c1-200
