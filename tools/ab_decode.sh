#!/bin/bash
# tools/ab_decode.sh <tag> <streams> -- <label> [VAR=value ...] [--bench-arg ...] -- <label> ... : the decode block of bench.py (HIP rows, 1 and 16 frame threads)
# once per setting, every run under its own timeout; one summary line per setting in gpurun_out/<tag>/summary.txt
TAG=$1; STREAMS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() {
  local label=$1; shift
  local envs=() args=()
  for a in "$@"; do if [[ "$a" == [A-Z_]*=* ]]; then envs+=("$a"); else args+=("$a"); fi; done
  ( export "${envs[@]}" OHEVC_NOOP=1; timeout 300 python bench.py --steps 3 --warmup 1 --no-kernels --no-cpu-baseline --no-frames --no-sizes --no-zscan --check-blocks 0 \
      --decode-hip-only --decode-streams $STREAMS "${args[@]}" > $OUT/$label.json 2> $OUT/$label.err; echo "rc $?" >> $OUT/$label.err )
  python - $label bench_detail.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.load(open(sys.argv[2]))["decode"]["streams"]
    row = []
    for name, s in d.items():
        for m in ("hip_1thread", "hip_16frame_threads"):
            r = s[m]
            row.append(f"{name[:6]}/{m[4:6]} {r['fps']:7.1f} ({r['fps_after_first_pass']:7.1f}) hook {r['per_picture']['frame_end_hook_ms']:.3f} up {r['per_picture']['upload_kib']}")
    print(f"{sys.argv[1]:18s}", " | ".join(row), "ok" if all(d[n]["bit_exact"] and d[n]["bit_exact_16_frame_threads"] for n in d) else "MISMATCH")
except Exception as e:
    print(f"{sys.argv[1]:18s} failed: {e}")
PY
  rm -f bench_detail.json
}
cur=()
for a in "$@"; do
  if [ "$a" = "--" ]; then [ ${#cur[@]} -gt 0 ] && run "${cur[@]}"; cur=(); else cur+=("$a"); fi
done
[ ${#cur[@]} -gt 0 ] && run "${cur[@]}"
exit 0
