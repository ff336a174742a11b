TAG=r4p; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/krows -o t -- python $ROOT/tools/kernel_rows.py > $ROOT/$OUT/kernel_rows.json 2> /tmp/krows.log ); tail -3 /tmp/krows.log
python tools/rocpd_summary.py stats /tmp/krows/t_results.db 2>/dev/null | cut -c1-170 | head -40 | tee $OUT/kernel_rows_rocprof_stats.txt
python - $OUT/kernel_rows.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(k, v if not isinstance(v, dict) else (v["kernel_ms"], v["frac"], v["checked"]))
PY
cat > /tmp/ipic.py <<'PY'
import sys, os, json, time
sys.path.insert(0, os.getcwd())
from oracle import pystream as ps
kw = dict(gop="random_access", nframes=2, seed=7, width=1920, height=1080, log2_ctb=6, init_qp=32,
          probs=dict(pred_mode=0.03, skip=0.55, merge_flag=0.7, split_cu=0.3, rqt_root_cbf=0.45, cbf_luma=0.5, cbf_chroma=0.25, split_transform=0.25, sig_coeff=0.35, last_x=0.5, last_y=0.5))
aus, _ = ps.generate(ps.StreamParams(**kw))
out = {}
for kind in ("c", "sse", "null", "hip"):
    best = None
    for _ in range(5):
        with ps.Decoder(kind, 1, 1) as d:
            t = time.perf_counter()
            d.L.ohdec_decode(d.h, aus[0], len(aus[0]), 1)
            dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    out[kind] = round(best * 1e3, 3)
print(json.dumps(dict(what="one 1080p intra picture of the encoder-like stream, one thread, ms from the call to the picture (first picture of a fresh decoder: includes the decoder's first-picture set-up)", **out)))
PY
python /tmp/ipic.py 2>/dev/null | grep '^{' | tee $OUT/intra_picture_ms.json
