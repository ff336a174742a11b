"""Regenerates tests/golden/shvc_streams.npz: synthetic two-layer (SHVC) streams + MD5 of every plane of BOTH layers as the UNTOUCHED
reference decoder outputs them (two decoders of oracle/_ref/libopenhevc_c.so opened the way gpac/modules/openhevc_dec/openHevcWrapper.c
does; the generator's own reconstruction and the reference's x86 / SSE4 decoder must agree first).

    make -C oracle && python tests/golden/make_shvc_streams.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pystream as ps          # noqa: E402
from shvc_cases import SHVC_CASES          # noqa: E402


def same(a, b):
    return len(a) == len(b) and all(np.array_equal(x, y) for fa, fb in zip(a, b) for x, y in zip(fa, fb))


out = {}
total = 0
for name, (kb, ke, pa) in SHVC_CASES.items():
    aus, gen_bl, gen_el = ps.generate_shvc(ps.StreamParams(**kb), ps.StreamParams(**ke), pa)
    ref_bl, ref_el = ps.decode_stream_shvc("c", aus)
    assert len(ref_bl) == len(ref_el) == kb["nframes"], (name, len(ref_bl), len(ref_el))
    assert same(gen_bl, ref_bl) and same(gen_el, ref_el), name
    if ps.have("sse"):
        sse_bl, sse_el = ps.decode_stream_shvc("sse", aus)
        assert same(sse_bl, ref_bl) and same(sse_el, ref_el), name
    out[name + ".data"] = np.frombuffer(b"".join(aus), dtype=np.uint8)
    out[name + ".sizes"] = np.array([len(a) for a in aus], dtype=np.int64)
    out[name + ".md5_bl"] = np.array([hashlib.md5(pl.tobytes()).hexdigest() for f in ref_bl for pl in f])
    out[name + ".md5_el"] = np.array([hashlib.md5(pl.tobytes()).hexdigest() for f in ref_el for pl in f])
    total += sum(len(a) for a in aus)
    print(f"{name:14s} {len(aus)} AUs {sum(len(a) for a in aus):7d} bytes  {ref_bl[0][0].shape[::-1]} -> {ref_el[0][0].shape[::-1]}")
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "shvc_streams.npz"), **out)
print("total stream bytes", total)
