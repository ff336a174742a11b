#!/bin/bash
# mc4 v2 (4 tiles in flight per wavefront, LDS tables): parity + A/B out of HBM + whole decoder A/B
TAG=${1:-r02zb}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_mc_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -6 ) 2>&1 | tee $OUT/pytest_mc.log
( OHEVC_MC_VARIANT=4 timeout 300 python -m pytest tests/test_ctx_gpu.py tests/test_stream_gpu.py tests/test_tables_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -4 ) 2>&1 | tee $OUT/pytest_decoder_mc4.log
for v in 4 3; do
  timeout 300 python tools/bench_kernels.py --resident --planes 8 --only mc --mc-variant $v 2>/dev/null | grep '^{' > $OUT/bench_mc_variant$v.jsonl
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/*/bench_mc_variant*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(d["kernel"][:90], round(d["ms"],4), round(d["frac_hbm_peak"],4))
PY
for v in 3 4; do
  OHEVC_MC_VARIANT=$v timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 --natural 2>/dev/null | tail -1 > $OUT/natural_mc$v.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/*/natural_mc*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d.get("bit_exact"), {k:(v.get("fps"), v.get("per_picture",{}).get("frame_end_hook_ms")) for k,v in d.items() if isinstance(v,dict) and "hip" in k})
PY
