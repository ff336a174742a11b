#!/bin/bash
# second sweep: the library's own asynchronous frame ends (issuer threads) against / with AMD_DIRECT_DISPATCH=0, three streams
TAG=${1:-knobs2}; OUT=gpurun_out/$TAG; mkdir -p $OUT
STREAMS=${STREAMS:-natural,intra_only,flat}
run() {
  local label=$1; shift
  ( export "$@" OHEVC_NOOP=1; timeout 200 python bench.py --steps 3 --warmup 1 --no-kernels --no-cpu-baseline --no-frames --no-sizes --no-zscan --check-blocks 0 \
      --decode-hip-only --decode-streams $STREAMS > $OUT/$label.json 2> $OUT/$label.err; echo "rc $?" >> $OUT/$label.err )
  python - $label bench_detail.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.load(open(sys.argv[2]))["decode"]["streams"]
    row = []
    for name, s in d.items():
        for m in ("hip_1thread", "hip_16frame_threads"):
            r = s[m]
            row.append(f"{name[:6]}/{m[4:6]} {r['fps']:7.1f} ({r['fps_after_first_pass']:7.1f}) hook {r['per_picture']['frame_end_hook_ms']:.3f}")
    print(f"{sys.argv[1]:28s}", " | ".join(row), "ok" if all(d[n]["bit_exact"] and d[n]["bit_exact_16_frame_threads"] for n in d) else "MISMATCH")
except Exception as e:
    print(f"{sys.argv[1]:28s} failed: {e}")
PY
  rm -f bench_detail.json
}
run baseline
run async4 OHHIP_ASYNC_ISSUE=1
run async1 OHHIP_ASYNC_ISSUE=1 OHEVC_ISSUER_THREADS=1
run async2 OHHIP_ASYNC_ISSUE=1 OHEVC_ISSUER_THREADS=2
run async8 OHHIP_ASYNC_ISSUE=1 OHEVC_ISSUER_THREADS=8
run dd0 AMD_DIRECT_DISPATCH=0
run dd0_async2 AMD_DIRECT_DISPATCH=0 OHHIP_ASYNC_ISSUE=1 OHEVC_ISSUER_THREADS=2
run baseline_again
run dd0_again AMD_DIRECT_DISPATCH=0
run async4_again OHHIP_ASYNC_ISSUE=1
