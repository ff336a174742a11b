# A/B of mc4q_kernel's aligned window loads (ohevc_debug_set_mc_variant(100 / 101))
for v in 101 100 101 100; do
python tools/bench_kernels.py --resident --planes 8 --only mc --mc-variant $v 2>/dev/null | grep '^{' | python -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['kernel'][:100], round(d['ms'],4), 'ms', round(d['frac_hbm_peak'],4))"
done
