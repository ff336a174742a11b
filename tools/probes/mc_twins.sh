# mc4q_kernel and its two twins (lab build): 104 = the kernel, 102 = its traffic alone, 103 = its arithmetic alone
export OHEVC_LAB_LIBRARY=1
for v in 104 102 103 104 102 103; do
python tools/bench_kernels.py --resident --planes 8 --only mc --mc-variant $v 2>/dev/null | grep '^{' | grep small-block | python -c '
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d["kernel"][:100], round(d["ms"], 4), "ms", round(d["frac_hbm_peak"], 4))'
done
