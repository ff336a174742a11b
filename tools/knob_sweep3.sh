#!/bin/bash
# parked frame ends (ohevc_frame_end_deferred) against the blocking frame end, all five 1080p streams, 1 and 16 frame threads; A/B/A/B
TAG=${1:-park}; OUT=gpurun_out/$TAG; mkdir -p $OUT
STREAMS=${STREAMS:-natural,flat,dense_qp22,intra_only,lowdelay_p}
run() {
  local label=$1; shift
  ( export "$@" OHEVC_NOOP=1; timeout 300 python bench.py --steps 3 --warmup 1 --no-kernels --no-cpu-baseline --no-frames --no-sizes --no-zscan --check-blocks 0 \
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
    print(f"{sys.argv[1]:18s}", " | ".join(row), "ok" if all(d[n]["bit_exact"] and d[n]["bit_exact_16_frame_threads"] for n in d) else "MISMATCH")
except Exception as e:
    print(f"{sys.argv[1]:18s} failed: {e}")
PY
  rm -f bench_detail.json
}
if [ "$2" = "threads" ]; then
  run block_1 OHHIP_PARK_FRAMES=0
  run park_t1 OHHIP_PARK_FRAMES=1 OHEVC_PARK_THREADS=1
  run park_t2 OHHIP_PARK_FRAMES=1 OHEVC_PARK_THREADS=2
  run block_2 OHHIP_PARK_FRAMES=0
  run park_t4 OHHIP_PARK_FRAMES=1 OHEVC_PARK_THREADS=4
  exit 0
fi
run block_1 OHHIP_PARK_FRAMES=0
run park_1 OHHIP_PARK_FRAMES=1
run block_2 OHHIP_PARK_FRAMES=0
run park_2 OHHIP_PARK_FRAMES=1
run park_dd0 OHHIP_PARK_FRAMES=1 AMD_DIRECT_DISPATCH=0
