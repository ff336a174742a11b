#!/bin/bash
# tools/knob_sweep.sh <tag> -- the HIP runtime's knobs against the frame end at 16 frame threads: the decode block of
# bench.py (HIP rows only, encoder-like + all-intra streams, 1 and 16 frame threads) once per setting, every run under its own timeout.
TAG=${1:-knobs}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() {   # <label> VAR=value ...
  local label=$1; shift
  ( export "$@" OHEVC_NOOP=1; timeout 150 python bench.py --steps 3 --warmup 1 --no-kernels --no-cpu-baseline --no-frames --no-sizes --no-zscan --check-blocks 0 \
      --decode-hip-only --decode-streams natural,intra_only > $OUT/$label.json 2> $OUT/$label.err; echo "rc $?" >> $OUT/$label.err )
  python - $label bench_detail.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.load(open(sys.argv[2]))["decode"]["streams"]
    row = []
    for name in ("natural", "intra_only"):
        s = d[name]
        for m in ("hip_1thread", "hip_16frame_threads"):
            r = s[m]
            row.append(f"{name[:6]}/{m[4:8]} {r['fps']:7.1f} ({r['fps_after_first_pass']:7.1f}) hook {r['per_picture']['frame_end_hook_ms']:.3f}")
    print(f"{sys.argv[1]:40s}", " | ".join(row), "ok" if all(d[n]["bit_exact"] and d[n]["bit_exact_16_frame_threads"] for n in d) else "MISMATCH")
except Exception as e:
    print(f"{sys.argv[1]:40s} failed: {e}")
PY
  rm -f bench_detail.json
}
run baseline
run direct_dispatch_0 AMD_DIRECT_DISPATCH=0
run signal_pool_1024 ROC_SIGNAL_POOL_SIZE=1024
run max_batch_64 DEBUG_CLR_MAX_BATCH_SIZE=64
run max_batch_10000 DEBUG_CLR_MAX_BATCH_SIZE=10000
run cpu_sync_0 DEBUG_CLR_BATCH_CPU_SYNC_SIZE=0
run cpu_sync_1000 DEBUG_CLR_BATCH_CPU_SYNC_SIZE=1000
run hw_queues_8 GPU_MAX_HW_QUEUES=8
run dev_kernarg HIP_FORCE_DEV_KERNARG=1
run active_wait_100us ROC_ACTIVE_WAIT_TIMEOUT=100
run cpu_wait_signal_0 ROC_CPU_WAIT_FOR_SIGNAL=0
run dynamic_queues DEBUG_HIP_DYNAMIC_QUEUES=1
run baseline_again
