TAG=${1:-r11}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 150 python tools/diag_cold_start.py 16 3 2> /dev/null | tail -1 > $OUT/cold_start_16_threads.json; cut -c1-1200 $OUT/cold_start_16_threads.json
