#!/bin/bash
# A/B of the chain kernel's slot forms ON ONE BOX: mean duration of the chain kernel (rocprofv3 kernel trace of the all-intra stream, one
# decoding thread), descriptors (0) against level records (4), interleaved
export TMPDIR=/tmp DIAG_GOP=intra DIAG_PASSES=2
for rep in 1 2 3; do for m in 0 4; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/ch$m -o t -- python $GRAFT_REPO_ROOT/tools/diag_overlap.py decode 1 natural chain_handover=$m > /tmp/ch$m.log 2>&1 )
  echo "handover $m: $(python tools/diag_overlap.py chain /tmp/ch$m/t_results.db | grep 'short const' | cut -c1-160) $(grep fps /tmp/ch$m.log | tail -1 | cut -c1-90)"
done; done
