#!/bin/bash
# tools/gpu.sh -- ONE parameterised script for every visit to the GPU box (it replaces the ~60 one-shot gpu_r0*.sh scripts of rounds 1-2).
#
#   gpurun --timeout 900 -- 'bash tools/gpu.sh <tag> -- <step> [args] -- <step> [args] ...'
#
# Every step writes under gpurun_out/<tag>/ (merged back by gpurun); the files worth keeping are copied to profiles/<tag>_<name> by hand
# (profiles/README.md maps file families to the steps below).  Steps:
#
#   suite [pytest args]          the driver's GPU tier: pytest tests -m gpu                          -> pytest_gpu.log
#   tests <files / -k ...>       selected test files                                                 -> pytest_<n>.log
#   smoke                        __graft_entry__.smoke()                                             -> smoke.log
#   bench [bench.py args]        one bench.py line                                                   -> bench<suffix>.json
#   bench_prof [bench.py args]   rocprofv3 --kernel-trace --stats of the bench command               -> kernel_stats.txt
#   rows_prof [only]             tools/kernel_rows.py under rocprofv3 --stats, then plain (HIP events)   -> kernel_rows_rocprof_stats.txt, kernel_rows_events.txt
#   pmc                          FETCH_SIZE / WRITE_SIZE passes of the bench command                 -> pmc_traffic.json
#   pmc_row <log2> <bd> <kernel> the same for bench.py --log2 <log2> --bit-depth <bd>                -> pmc_traffic_log2_<log2>_<bd>bit.json
#   counters <kernel> <cmd ...>  SQ / TCP / TCC counter passes of <cmd>, rows of kernels matching    -> counters_<kernel>.txt
#   kernels <only> [args]        tools/bench_kernels.py --resident --planes 8 --only <only>          -> bench_kernels_<only>.jsonl
#   kernels_prof <only> [args]   the same command under rocprofv3 (stats + the two PMC passes)       -> kernel_stats_<only>_<n>.txt, pmc_<only>_<n>.jsonl
#   decode <bench_decode args>   tools/bench_decode.py (whole decoder, all thread modes)             -> decode_<n>.json
#   chain [flat|natural]         per-launch durations and stream-idle gaps of one decoding thread    -> chain.jsonl, overlap.json
#   overlap <threads>            kernel overlap between frame threads                                -> overlap_<threads>.jsonl
#   timing <threads> [args]      OHEVC_TRACE=timing split of the frame-end hook                      -> timing_<threads>.txt
#   fuzz <seconds> [seed]        tools/fuzz_streams.py on the device                                 -> fuzz.json
#   frames <ranks> [args]        bench.py --mode frames (ranks > 1: --frames-one-gpu through gloo)   -> frames_<ranks>.json
#   ab <variants> [size]         tools/ab_tu_variants.py (lab build)                                 -> ab_tu_variants.txt
#   probe                        tools/hbm_probe + tools/probes/dispatch_probe                       -> hbm_probe.jsonl, dispatch_probe.jsonl
#   sh <command ...>             anything else, output kept                                          -> sh_<n>.log
TAG=${1:-visit}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids'
n=0

prof_stats() {      # <dir> <name>: rocprofv3 --kernel-trace --stats summary of "$@" (after the two fixed arguments)
  local dir=$1 name=$2; shift 2
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $dir -o t -- "$@" > /tmp/prof_$name.log 2>&1 )
  # the header, every row of the product's kernels (ohevc::), and the first few rows of everything else (torch's fill kernels used to crowd
  # the product's rows out of a 14-line head: VERDICT r5 weak 8)
  python tools/rocpd_summary.py stats $dir/t_results.db 2>/dev/null | cut -c1-170 | awk -v other=${STATS_ROWS:-6} 'NR == 1 || /ohevc::/ { print; next } other-- > 0 { print }'
}

# a step may be preceded by VAR=value words: they are exported for that step only (and become part of its file names)
step() {
  n=$((n + 1))
  local envs=()
  while [[ "$1" == [A-Za-z_]*=* ]]; do envs+=("$1"); shift; done
  if [ ${#envs[@]} -gt 0 ]; then
    ( export "${envs[@]}"; ENVTAG=$(echo "${envs[*]}" | tr -c 'a-zA-Z0-9\n' '_'); run_step "$@" )
  else
    ENVTAG=""; run_step "$@"
  fi
}

run_step() {
  local s=$1; shift
  echo "== [$TAG] ${ENVTAG:+($ENVTAG) }$s $*"
  case $s in
    suite)
      ( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x "$@" 2>&1 | grep -v "$NOISE" | tail -12 ) 2>&1 | cut -c1-400 | tee $OUT/pytest_gpu.log ;;
    tests)
      ( time timeout 900 python -m pytest "$@" -q -p no:cacheprovider -x 2>&1 | grep -v "$NOISE" | tail -30 ) 2>&1 | cut -c1-400 | tee $OUT/pytest_$n.log ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$NOISE" | tail -3 | tee $OUT/smoke.log ;;
    bench)
      local sfx=$(echo "$ENVTAG $*" | tr -c 'a-zA-Z0-9\n' '_' | sed 's/__*/_/g; s/^_//; s/_$//')
      # the last stdout line = the compact line the driver parses (bench.compact_line); the full object = bench_detail.json
      timeout 900 python bench.py "$@" 2> $OUT/bench_${sfx:-default}.err | tail -1 > $OUT/bench${sfx:+_$sfx}.json
      [ -f bench_detail.json ] && cp bench_detail.json $OUT/bench${sfx:+_$sfx}_detail.json
      echo "  line: $(wc -c < $OUT/bench${sfx:+_$sfx}.json) bytes"; cut -c1-1200 $OUT/bench${sfx:+_$sfx}.json; tail -3 $OUT/bench_${sfx:-default}.err | grep -v "$NOISE"
      python - $OUT/bench${sfx:+_$sfx}_detail.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("  frac", d["roofline"]["frac"], "zscan", d.get("zscan", {}).get("frac"), "checked", d.get("checked"))
    for k, v in d.get("decode", {}).get("streams", {}).items():
        print("  decode", k, {kk: (vv.get("fps"), vv.get("per_picture", {}).get("frame_end_hook_ms"), vv.get("per_picture", {}).get("launches")) for kk, vv in v.items() if isinstance(vv, dict)})
except Exception as e:
    print("  (no summary:", e, ")")
PY
      ;;
    bench_prof)
      prof_stats /tmp/prof_bench bench python $ROOT/bench.py --no-cpu-baseline --no-decode --no-zscan "$@" | tee $OUT/kernel_stats.txt ;;
    rows_prof)      # every row of the per-kernel table under rocprofv3: the per-kernel average durations the bench's HIP-event times are to agree with
      STATS_ROWS=0 prof_stats /tmp/prof_rows rows python $ROOT/tools/kernel_rows.py "$@" | tee $OUT/kernel_rows_rocprof_stats.txt
      timeout 600 python tools/kernel_rows.py "$@" > $OUT/kernel_rows.json 2>/dev/null
      python - $OUT/kernel_rows.json <<'PY' | tee $OUT/kernel_rows_events.txt
import json, sys
for k, v in json.load(open(sys.argv[1])).items():
    if isinstance(v, dict): print(f"{k:52s} {1e3 * v['kernel_ms']:9.2f} us  frac {v['frac']:.3f} checked {v['checked']}")
    else: print(k, v)
PY
      ;;
    pmc)
      local CMD="python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-decode --no-zscan --check-blocks 0"
      ( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_fetch -o p -- $CMD > /tmp/pmc_fetch.log 2>&1
        timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_write -o p -- $CMD > /tmp/pmc_write.log 2>&1 )
      python tools/pmc_traffic.py /tmp/pmc_fetch/p_results.db /tmp/pmc_write/p_results.db "tu_idct32_tile1_kernel<unsigned char" | tee $OUT/pmc_traffic.json ;;
    pmc_row)      # <log2> <bit depth> <kernel name prefix>: the same two passes for another row of the residual kernels (bench.py --log2 / --bit-depth)
      local l2=$1 bdp=$2 kn=$3
      local CMD="python $ROOT/bench.py --log2 $l2 --bit-depth $bdp --steps 4 --warmup 2 --no-cpu-baseline --no-decode --no-kernels --no-frames --no-zscan --check-blocks 0"
      ( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmcr_fetch_${l2}_$bdp -o p -- $CMD > /dev/null 2>&1
        timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmcr_write_${l2}_$bdp -o p -- $CMD > /dev/null 2>&1 )
      python tools/pmc_traffic.py /tmp/pmcr_fetch_${l2}_$bdp/p_results.db /tmp/pmcr_write_${l2}_$bdp/p_results.db "$kn" | tee $OUT/pmc_traffic_log2_${l2}_${bdp}bit.json ;;
    counters)
      local k=$1 i=0; shift
      # (at most four counters per pass: larger sets were refused by the counter scheduler and left the files empty)
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
                 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM" \
                 "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_READ_sum" \
                 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
        i=$((i + 1))
        ( cd $ROOT && timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/ctr_$i -o p -- "$@" > /tmp/ctr_$i.log 2>&1 ) || tail -3 /tmp/ctr_$i.log
        python tools/rocpd_summary.py pmc /tmp/ctr_$i/p_results.db $k 2>&1 | cut -c1-220 | tee -a $OUT/counters_$k.txt
      done ;;
    kernels)
      local only=$1; shift
      timeout 600 python tools/bench_kernels.py --resident --planes 8 --only $only "$@" 2>/dev/null | grep '^{' | tee -a $OUT/bench_kernels_$only.jsonl | \
        python -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['kernel'][:90], round(d['ms'],4), 'ms', round(d.get('alg_GBps',0),1), 'GB/s', round(d['frac_hbm_peak'],4))" ;;
    kernels_prof)
      local only=$1; shift
      local CMD="python $ROOT/tools/bench_kernels.py --resident --planes 8 --only $only $*"
      prof_stats /tmp/kp_$only kp_$only $CMD | tee $OUT/kernel_stats_${only}_$n.txt
      ( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/kpf_$only -o p -- $CMD > /dev/null 2>&1
        timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/kpw_$only -o p -- $CMD > /dev/null 2>&1 )
      python tools/pmc_per_kernel.py /tmp/kp_$only/t_results.db /tmp/kpf_$only/p_results.db /tmp/kpw_$only/p_results.db | tee $OUT/pmc_${only}_$n.jsonl | cut -c1-300 ;;
    decode)
      timeout 900 python tools/bench_decode.py "$@" 2>/dev/null | tail -1 > $OUT/decode_$n${ENVTAG:+_$ENVTAG}.json; [ -n "$ENVTAG" ] && cp $OUT/decode_$n${ENVTAG:+_$ENVTAG}.json $OUT/decode_$n.json
      python - $OUT/decode_$n.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d.get("workload"), "bit_exact", d.get("bit_exact"), d.get("bit_exact_frame_threads"))
for k, v in d.items():
    if isinstance(v, dict) and "fps" in v:
        pp = v.get("per_picture", {})
        print(" ", k, v["fps"], "fps", pp.get("frame_end_hook_ms", ""), pp.get("launches", ""))
PY
      ;;
    chain)
      ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/ch -o t -- python $ROOT/tools/diag_overlap.py decode 1 ${1:-} > $ROOT/$OUT/chain_decode.log 2>&1 )
      tail -1 $OUT/chain_decode.log | cut -c1-300
      python tools/diag_overlap.py analyze /tmp/ch/t_results.db | tee $OUT/overlap.json | cut -c1-600
      python tools/diag_overlap.py chain /tmp/ch/t_results.db | tee $OUT/chain.jsonl | cut -c1-400 ;;
    overlap)
      local th=${1:-16}
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/ov$th -o t -- python $ROOT/tools/diag_overlap.py decode $th > /tmp/ov$th.log 2>&1 )
      tail -1 /tmp/ov$th.log | tee $OUT/overlap_$th.jsonl | cut -c1-300
      python tools/diag_overlap.py analyze /tmp/ov$th/t_results.db | tee -a $OUT/overlap_$th.jsonl | cut -c1-600 ;;
    timing)
      local th=${1:-16}; shift
      OHEVC_TRACE=timing timeout 300 python tools/diag_overlap.py decode $th "$@" 2>&1 | grep -v "$NOISE" | grep "timing:\|fps" | sort | uniq -c | sort -rn | head -40 | cut -c1-300 | tee $OUT/timing_$th.txt ;;
    fuzz)
      ( timeout $((${1:-60} + 60)) python tools/fuzz_streams.py ${1:-60} ${2:-$RANDOM} 2>&1 | grep -v "$NOISE" | tail -2 ) | tee $OUT/fuzz.json | cut -c1-600 ;;
    frames)
      local r=${1:-1}; shift
      if [ $r -le 1 ]; then
        timeout 400 python bench.py --mode frames --steps 2 --warmup 1 "$@" 2>/dev/null | tail -1 | tee $OUT/frames_1.json | cut -c1-600
      else
        ( cd /tmp && timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $r --master-addr 127.0.0.1 --master-port 2964$r $ROOT/bench.py --gpus $r --mode frames \
            --frames-one-gpu --steps 2 --warmup 1 "$@" 2>/dev/null | tail -1 ) | tee $OUT/frames_$r.json | cut -c1-600
      fi ;;
    ab)
      timeout 600 python tools/ab_tu_variants.py "$@" 2>&1 | tail -6 | tee $OUT/ab_tu_variants.txt ;;
    probe)
      [ -x tools/hbm_probe ] && timeout 300 tools/hbm_probe 2 2>&1 | tee $OUT/hbm_probe.jsonl | cut -c1-300
      [ -x tools/probes/dispatch_probe ] && timeout 300 tools/probes/dispatch_probe "$@" 2>&1 | tee $OUT/dispatch_probe.jsonl | cut -c1-300 ;;
    sh)
      ( timeout 900 "$@" 2>&1 | grep -v "$NOISE" | tail -40 ) > $OUT/sh_$n.log; cut -c1-600 $OUT/sh_$n.log ;;
    *)
      echo "unknown step $s" ;;
  esac
}

args=()
for a in "$@"; do
  if [ "$a" = "--" ]; then
    [ ${#args[@]} -gt 0 ] && step "${args[@]}"
    args=()
  else
    args+=("$a")
  fi
done
[ ${#args[@]} -gt 0 ] && step "${args[@]}"
nproc > $OUT/host_cores.txt
find $OUT /tmp -maxdepth 3 -name '*.db' -size +20M -delete 2>/dev/null
exit 0
