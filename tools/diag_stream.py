"""GPU diagnostic: synthetic streams through the reference decoder with CPU tables vs HIP tables (integration/hip_hooks.c)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pystream as ps

CASES = [
    dict(gop="intra", nframes=2, seed=1),
    dict(gop="lowdelay_p", nframes=4, seed=2),
    dict(gop="lowdelay_b", nframes=5, seed=3),
    dict(gop="random_access", nframes=9, seed=4, width=416, height=240, log2_ctb=6),
    dict(gop="random_access", nframes=9, seed=5, bit_depth=10, width=208, height=120, log2_ctb=6),
    dict(gop="lowdelay_b", nframes=4, seed=6, cu_qp_delta_depth=1, weighted_pred=1, weighted_bipred=1),
    dict(gop="lowdelay_b", nframes=4, seed=7, pcm=7),
    dict(gop="lowdelay_b", nframes=4, seed=8, constrained_intra=1),
    dict(gop="lowdelay_b", nframes=4, seed=9, wpp=1),
    dict(gop="lowdelay_b", nframes=4, seed=10, tiles=(2, 2)),
    dict(gop="lowdelay_b", nframes=4, seed=11, slices_per_picture=3),
    dict(gop="lowdelay_b", nframes=4, seed=12, rext=1),
]

def first_diff(a, b):
    d = np.argwhere(a != b)
    return None if d.size == 0 else (tuple(int(v) for v in d[0]), int(a[tuple(d[0])]), int(b[tuple(d[0])]), len(d))

if len(sys.argv) > 1:
    import json
    CASES = [json.loads(a) for a in sys.argv[1:]]
    for c in CASES:
        if "tiles" in c:
            c["tiles"] = tuple(c["tiles"])
bad = 0
for kw in CASES:
    p = ps.StreamParams(**kw)
    aus, gen_frames = ps.generate(p)
    ref = ps.decode_stream("c", aus)
    t = time.time()
    try:
        hip = ps.decode_stream("hip", aus, int(os.environ.get("DIAG_THREADS", "1")), 1)
    except Exception as e:
        print("HIP FAIL", kw, e); bad += 1; continue
    dt = time.time() - t
    ok = len(ref) == len(hip)
    msg = ""
    for i, (fa, fb) in enumerate(zip(ref, hip)):
        for c in range(3):
            fd = first_diff(fa[c], fb[c])
            if fd:
                ok = False
                msg += f"\n    [frame {i} plane {c}: first (y,x)={fd[0]} ref={fd[1]} hip={fd[2]} ndiff={fd[3]}]"
                if os.environ.get("DIAG_VERBOSE"):
                    for (y, x) in np.argwhere(fa[c] != fb[c])[:4]:
                        y0, x0 = max(0, y - 2), max(0, x - 2)
                        msg += f"\n      at (y={y},x={x}) ref:\n{fa[c][y0:y+3, x0:x+3]}\n      hip:\n{fb[c][y0:y+3, x0:x+3]}"
    print("OK " if ok else "MISMATCH", kw, f"frames {len(ref)}/{len(hip)} {dt:.2f}s", msg)
    bad += not ok
print("bad", bad)
sys.exit(1 if bad else 0)
