#!/bin/bash
# frame threads vs the runtime's queue knobs: GPU_MAX_HW_QUEUES (default 4), HIP_FORCE_DEV_KERNARG
TAG=${1:-r02hq}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
nproc > $OUT/nproc.txt
run() { # name, env...
  name=$1; shift
  env "$@" timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 2>/dev/null | tail -1 > $OUT/flat_$name.json
}
run default A=1
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq16 GPU_MAX_HW_QUEUES=16
run devkernarg HIP_FORCE_DEV_KERNARG=1
run hwq16_devkernarg GPU_MAX_HW_QUEUES=16 HIP_FORCE_DEV_KERNARG=1
python - <<PY
import json,glob
print("cores", open("$OUT/nproc.txt").read().strip())
for f in sorted(glob.glob("$OUT/flat_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d.get("bit_exact"), d.get("bit_exact_frame_threads"), {k.replace("hip_backend","hb"):(v.get("fps"), v.get("per_picture",{}).get("frame_end_hook_ms")) for k,v in d.items() if isinstance(v,dict) and "hip" in k})
PY
