# device pictures in batches: the first pass again, and the stream tests that allocate the most (parameter-set changes, 4K / 8K, two layers, decoders opened and closed)
TAG=${1:-r15}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
NOISE='^\[MD5\|^POC\|^[0-9a-f]\{32\}$\|^\]$\|The cu_qp_delta\|PPS extension\|partially impl\|amdgpu.ids\|IRAP'
timeout 100 python tools/diag_cold_start.py 16 3 2> /dev/null | tail -1 > $OUT/cold_start_16_threads_picture_batches.json; cut -c1-900 $OUT/cold_start_16_threads_picture_batches.json
timeout 200 python -m pytest tests/test_stream_gpu.py tests/test_shvc_stream_gpu.py tests/test_ctx_gpu.py -q -p no:cacheprovider -k "frame_threads_share or parameter_sets_change or fifty_decoders or two_decoders or damaged or config or shvc_both_layers or pairs or ctx" 2>&1 | grep -v "$NOISE" | tail -3 | cut -c1-300 | tee $OUT/pytest_subset.log
