# what a workgroup that spans two CTBs would see: SAO blocks of 128 x 64 samples against 64 x 64 (8 bit: 128- against 64-byte rows)
for w in 64 128 64 128; do for k in 1 2; do
python tools/bench_kernels.py --resident --planes 8 --only sao --sao-width $w --sao-class $k 2>/dev/null | grep '^{' | python -c '
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d["kernel"][:70], round(d["ms"], 4), "ms", round(d["frac_hbm_peak"], 4))'
done; done
