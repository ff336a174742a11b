TAG=${1:-r12}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 150 python tools/diag_first_pass.py 16 2> /dev/null | tail -1 > $OUT/first_pass_16_threads.json; cut -c1-2500 $OUT/first_pass_16_threads.json
