# where a decoding thread's time goes on the all-intra stream: the hooks recording without a device (OHHIP_RECORD_ONLY=1) against the full back end
for e in "OHHIP_RECORD_ONLY=1" "" "OHHIP_RECORD_ONLY=1" ""; do
  for th in 16 1; do
    echo "== $e threads $th"
    env $e DIAG_GOP=intra DIAG_NATURAL=1 timeout 300 python tools/diag_overlap.py decode $th 2>&1 | grep '"fps"'
  done
done
