"""GPU diagnostic: decode one synthetic stream (JSON params in argv[1]) with C tables and HIP tables, print a map of where and by how much they differ."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pystream as ps

kw = json.loads(sys.argv[1])
if "tiles" in kw:
    kw["tiles"] = tuple(kw["tiles"])
aus, _ = ps.generate(ps.StreamParams(**kw))
th, tt = int(os.environ.get("DIAG_THREADS", "1")), int(os.environ.get("DIAG_TYPE", "1"))
ref = ps.decode_stream("c", aus, th if tt == 2 else 1, tt if tt == 2 else 1)
hip = ps.decode_stream("hip", aus, th, tt)
for i, (fa, fb) in enumerate(zip(ref, hip)):
    for c in range(3):
        d = fb[c].astype(np.int64) - fa[c].astype(np.int64)
        if not d.any():
            continue
        ys, xs = np.nonzero(d)
        print(f"frame {i} plane {c}: {len(ys)} diffs, bbox x {xs.min()}..{xs.max()} y {ys.min()}..{ys.max()}, values {np.unique(d[d != 0])[:12]}")
        y0, x0 = (ys.min() // 8) * 8, (xs.min() // 8) * 8
        for y in range(y0, min(y0 + 40, d.shape[0]), 4):
            print("   y=%4d " % y + " ".join("%5d" % d[y, x] for x in range(x0, min(x0 + 48, d.shape[1]), 4)))
        print("   ref row %d: %s" % (ys.min(), fa[c][ys.min(), x0:x0 + 40:4]))
        print("   hip row %d: %s" % (ys.min(), fb[c][ys.min(), x0:x0 + 40:4]))
        break
    else:
        continue
    break
