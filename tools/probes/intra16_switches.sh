# the all-intra stream at 16 frame threads under the hooks' two pipeline switches
for e in "" "OHHIP_DEFER_DOWNLOAD=1" "OHHIP_ASYNC_ISSUE=1" "OHHIP_DEFER_DOWNLOAD=1 OHHIP_ASYNC_ISSUE=1" "" "OHHIP_DEFER_DOWNLOAD=1"; do
  echo "== $e"
  env $e DIAG_GOP=intra DIAG_NATURAL=1 timeout 300 python tools/diag_overlap.py decode 16 2>&1 | grep '"fps"'
  env $e DIAG_NATURAL=1 timeout 300 python tools/diag_overlap.py decode 16 2>&1 | grep '"fps"'
done
