TAG=${1:-r13}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
DIAG_DUMP=1 timeout 150 python tools/diag_first_pass.py 16 2> /dev/null | tail -1 > $OUT/first_pass_16_threads_pictures.json; python -c "
import json; d=json.load(open('$OUT/first_pass_16_threads_pictures.json'))
print(d['first_pass']['span_ms'], d['second_pass']['span_ms'])
for r in d['first_pass_pictures']: print(r)"
