"""Decode one synthetic stream with the HIP-backed reference decoder (profiling target for rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pystream as ps
size = sys.argv[1] if len(sys.argv) > 1 else "1920x1080"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 17
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 1
bd = int(sys.argv[4]) if len(sys.argv) > 4 else 8
w, h = map(int, size.split("x"))
aus, _ = ps.generate(ps.StreamParams(gop="random_access", nframes=n, seed=7, width=w, height=(h + 7) // 8 * 8, log2_ctb=6, bit_depth=bd))
for rep in range(2):
    t = time.perf_counter()
    out = ps.decode_stream("hip", aus, threads, 1)
    print(f"{size} x{n} threads {threads}: {len(out) / (time.perf_counter() - t):.1f} fps")
