#!/bin/bash
# tools/ab_round6.sh <tag> -- A/B runs of round 6's switches on the five 1080p streams of bench.py's decode block (HIP rows only)
TAG=${1:-ab6}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
one() {   # <name> [VAR=value ...] -- [bench args]
  local name=$1; shift; local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  ( export "${envs[@]}" DUMMY_=1; timeout 600 python bench.py --no-cpu-baseline --no-kernels --no-frames --no-zscan --decode-hip-only --no-sizes --steps 2 --warmup 1 --check-blocks 0 "$@" > $OUT/$name.out 2> /dev/null )
  python - $name <<'PY'
import json, sys
try:
    d = json.load(open("bench_detail.json"))
    row = []
    for k, v in d["decode"]["streams"].items():
        a, b = v["hip_1thread"], v["hip_16frame_threads"]
        row.append(f"{k[:6]} {a['fps']:6.1f} ({a['fps_after_first_pass']:6.1f}) | {b['fps']:6.1f} ({b['fps_after_first_pass']:6.1f}) hook {b['per_picture']['frame_end_hook_ms']:.2f}")
    print(f"{sys.argv[1]:26s}", " || ".join(row), "ok" if d["decode"]["bit_exact"] else "MISMATCH")
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
}
if [ -n "$AB_ONLY" ]; then for v in $AB_ONLY; do case $v in
    baseline*) one $v -- $AB_ARGS ;;
    fetch_issues_copies) one $v OHHIP_QUEUE_DOWNLOAD=0 -- ;;
    three_priority_pools) one $v -- --debug-set long_chain_pools=4 ;;
    decoder_frames_pinned) one $v OHHIP_OWN_FRAMES=0 -- ;;
    flush_*) one $v OHHIP_FLUSH_INTRA_KIB=${v#flush_} -- $AB_ARGS ;;
    *) one $v -- --debug-set $v ;;
  esac; done; exit 0; fi
one baseline              --
one fetch_issues_copies   OHHIP_QUEUE_DOWNLOAD=0 --
one three_priority_pools  -- --debug-set long_chain_pools=4
one decoder_frames_pinned OHHIP_OWN_FRAMES=0 --
one baseline_again        --
exit 0
