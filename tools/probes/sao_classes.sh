for v in 0; do for k in 0 1 2 3; do
python tools/bench_kernels.py --resident --planes 8 --only sao --sao-variant $v --sao-class $k 2>/dev/null | grep '^{' | python -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['kernel'][:60], round(d['ms'],4), 'ms', round(d['frac_hbm_peak'],4))" | grep edge
done; done
