TAG=r4n; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids'
( time timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_ctx_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v "$NOISE" | tail -6 ) 2>&1 | cut -c1-300 | tee $OUT/pytest_flush.log
for cfg in "32 2 1500" "32 2 0" "32 2 600" "32 2 4000"; do set -- $cfg
  OHEVC_INTRA_CHAIN_WAVES=$1 OHEVC_INTRA_CHAIN_MIN_RUN=$2 OHHIP_FLUSH_INTRA_JOBS=$3 timeout 900 python bench.py --no-kernels --no-cpu-baseline --no-zscan --steps 20 --decode-hip-only 2> $OUT/bench.err | tail -1 > $OUT/bench_$1_$2_$3.json
  python - $OUT/bench_$1_$2_$3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.get("decode", {}).get("streams", {}).items():
    print("  decode", sys.argv[1].split("bench_")[1][:-5], k, {kk: (vv.get("fps"), vv.get("per_picture", {}).get("frame_end_hook_ms"), vv.get("per_picture", {}).get("launches")) for kk, vv in v.items() if isinstance(vv, dict)})
PY
done
rm -f /tmp/ft.txt
OHHIP_TRACE_FRAMES=/tmp/ft.txt python tools/diag_overlap.py decode 16 natural 2>/dev/null | grep fps | tee $OUT/frames_16_natural.jsonl
python tools/frame_trace.py /tmp/ft.txt | tee -a $OUT/frames_16_natural.jsonl
gzip -c /tmp/ft.txt > $OUT/frame_trace_16_natural.txt.gz
cat > /tmp/dec8k.py <<'PY'
import sys, os, json, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from oracle import pystream as ps
import bench_decode as bd
size = sys.argv[1]; w, h = map(int, size.split("x")); frames = int(sys.argv[2]); bdp = int(sys.argv[3])
kw = dict(gop="random_access", nframes=frames, seed=7, width=w, height=(h + 7) // 8 * 8, log2_ctb=6, bit_depth=bdp, init_qp=32,
          probs=dict(pred_mode=0.03, skip=0.55, merge_flag=0.7, split_cu=0.3, rqt_root_cbf=0.45, cbf_luma=0.5, cbf_chroma=0.25, split_transform=0.25, sig_coeff=0.35, last_x=0.5, last_y=0.5))
aus, _ = ps.generate(ps.StreamParams(**kw))
bd.PASSES = 2
L = ps._load("hip"); sec = C.c_double(); cnt = (C.c_longlong * 8)()
out = {}
for th in (1, 8):
    L.ohdec_backend_profile(C.byref(sec), cnt)
    dt, n = bd.timed_decode("hip", aus, th, 1)
    L.ohdec_backend_profile(C.byref(sec), cnt)
    out[f"hip_{th}"] = (round(n / dt, 2), round(1e3 * sec.value / max(1, cnt[0]), 3), round(cnt[1] / max(1, cnt[0]), 1))
print(json.dumps(dict(size=size, env={k: v for k, v in os.environ.items() if k.startswith("OHEVC_INTRA") or k.startswith("OHHIP_FLUSH")}, **out)))
PY
for cfg in "32 2 1500" "32 2 0"; do set -- $cfg
  OHEVC_INTRA_CHAIN_WAVES=$1 OHEVC_INTRA_CHAIN_MIN_RUN=$2 OHHIP_FLUSH_INTRA_JOBS=$3 timeout 600 python /tmp/dec8k.py 7680x4320 5 10 2>/dev/null | grep '^{' | tee -a $OUT/decode_8k_ab.jsonl
done
