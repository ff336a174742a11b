# page locks taken at the frame end instead of in the picture's serial prologue: the first pass again, and the stream tests that lean on page locks
TAG=${1:-r14}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids\|IRAP'
timeout 100 python tools/diag_cold_start.py 16 3 2> /dev/null | tail -1 > $OUT/cold_start_16_threads_pins_at_frame_end.json; cut -c1-900 $OUT/cold_start_16_threads_pins_at_frame_end.json
timeout 200 python -m pytest tests/test_stream_gpu.py -q -p no:cacheprovider -k "frame_threads_share or parameter_sets_change or fifty_decoders or two_decoders or damaged or config3 or pipelined" 2>&1 | grep -v "$NOISE" | tail -3 | cut -c1-300 | tee $OUT/pytest_streams_subset.log
