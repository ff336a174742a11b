#!/bin/bash
# One short GPU-box visit during development: parity tests of the files given in $2.., then a fuzz run.   bash tools/gpu_visit.sh <tag> <fuzz seconds> [pytest args]
TAG=${1:-visit}; FUZZ=${2:-60}; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest "$@" -q -p no:cacheprovider 2>&1 | tail -12 | cut -c1-600 | tee $OUT/pytest.log
timeout $((FUZZ + 60)) python tools/fuzz_streams.py $FUZZ $RANDOM 2>&1 | grep -v "The cu_qp_delta\|PPS extension\|partially impl" | tail -6 | cut -c1-1800 | tee $OUT/fuzz.log
if [ -f tools/diag_cases.txt ]; then
  while read -r line; do
    for t in 1 3 3; do DIAG_THREADS=$t timeout 120 python tools/diag_stream.py "$line" 2>&1 | grep -v "The cu_qp_delta\|PPS extension\|partially impl" | tail -4 | cut -c1-700; done
  done < tools/diag_cases.txt | tee $OUT/diag.log
fi
