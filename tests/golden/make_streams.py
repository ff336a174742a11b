"""Regenerates tests/golden/streams.npz: synthetic HEVC streams + MD5 of every plane the UNTOUCHED reference decoder
(oracle/_ref/libopenhevc_c.so, built in place from /root/reference) outputs for them.

    make -C oracle && python tests/golden/make_streams.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pystream as ps          # noqa: E402
from stream_cases import CASES             # noqa: E402

out = {}
total = 0
for name, kw in CASES.items():
    aus, gen_frames = ps.generate(ps.StreamParams(**kw))
    ref = ps.decode_stream("c", aus)
    assert len(ref) == len(gen_frames) == kw["nframes"], (name, len(ref), len(gen_frames))
    for a, b in zip(gen_frames, ref):
        for x, y in zip(a, b):
            assert np.array_equal(x, y), name
    md5 = [hashlib.md5(pl.tobytes()).hexdigest() for f in ref for pl in f]
    out[name + ".data"] = np.frombuffer(b"".join(aus), dtype=np.uint8)
    out[name + ".sizes"] = np.array([len(a) for a in aus], dtype=np.int64)
    out[name + ".md5"] = np.array(md5)
    total += sum(len(a) for a in aus)
    print(f"{name:18s} {len(aus)} AUs {sum(len(a) for a in aus):7d} bytes")
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "streams.npz"), **out)
print("total stream bytes", total)
