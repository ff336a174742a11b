#!/bin/bash
# the wait for the reference pictures' frame ends moved behind the staging copy + upload: thread-mode parity on the device, timing split
# (OHEVC_TRACE_TIMING), whole decoder flat / natural
TAG=${1:-r02tt2}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_stream_gpu.py tests/test_ctx_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v '^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$' | tail -3 ) 2>&1 | tee $OUT/pytest.log
OHEVC_TRACE_TIMING=1 timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 2> $OUT/flat_stderr.txt | tail -1 > $OUT/flat.json
timeout 150 python tools/bench_decode.py --size 1920x1080 --frames 33 --natural 2>/dev/null | tail -1 > $OUT/natural.json
( timeout 120 python tools/fuzz_streams.py 60 777 2>&1 | tail -1 ) | tee $OUT/fuzz.json | cut -c1-300
grep "^timing" $OUT/flat_stderr.txt | python3 -c "
import sys,re,collections
g=collections.defaultdict(list)
for l in sys.stdin:
    m=re.search(r'(\d+) frames: frame_end ([\d.]+) ms/frame of which waiting for reference frames ([\d.]+)',l)
    if m: g['1thread' if int(m.group(1))==33 else 'threads'].append((int(m.group(1)),float(m.group(2)),float(m.group(3))))
for k,v in g.items():
    tot=sum(n for n,_,_ in v); print(k,'frames',tot,'frame_end ms/frame %.3f'%(sum(n*a for n,a,_ in v)/tot),'of which waiting for refs %.3f'%(sum(n*b for n,_,b in v)/tot))
" | tee $OUT/timing_split.txt
python - <<PY
import json,glob
for f in ["$OUT/flat.json","$OUT/natural.json"]:
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d.get("bit_exact"), d.get("bit_exact_frame_threads"), {k.replace("hip_backend","hb"):(v.get("fps"), v.get("per_picture",{}).get("frame_end_hook_ms")) for k,v in d.items() if isinstance(v,dict) and "hip" in k})
PY
