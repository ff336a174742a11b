#!/usr/bin/env python3
"""Which of the two round-5 switches a frame-threaded mismatch belongs to: golden streams through the hooked decoder with the long-chain stream
forced on every picture that has intra levels / never, and the coefficient upload compact / whole, each combination `reps` times.
    python tools/diag_switches.py [threads] [reps] [stream ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pystream as ps                       # noqa: E402
from test_stream_cpu import frames_md5, load_golden     # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
names = sys.argv[3:] or ["cross_444_10b_tqb", "ra_8b_ctb64", "fmt444_14b_cip_cross", "ldb_10b", "tiles", "pcm"]
product = ps._product_lib()
for levels, compact in ((0, 1), (1, 0), (1, 1), (0, 0)):
    product.ohevc_debug_set_long_chain_levels(levels)
    product.ohevc_debug_set_compact_coeffs(compact)
    for name in names:
        aus, md5 = load_golden(name)
        bad = sum(frames_md5(ps.decode_stream("hip", aus, threads, 1)) != md5 for _ in range(reps))
        print(f"long_chain_levels {levels} compact {compact} {name}: {bad} / {reps} runs differ", flush=True)
