#!/usr/bin/env python3
"""One 1080p intra picture of the encoder-like stream, one decoding thread, milliseconds from the call to the picture: (a) the first picture of
a fresh decoder (includes the decoder's first-picture set-up; what profiles/r4p_intra_picture_ms.json measured), (b) the intra picture that
opens the SECOND pass of the stream through the same decoder (steady state).  Best of 5 decoders each.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pystream as ps       # noqa: E402
import bench                             # noqa: E402

kw = dict(gop="random_access", nframes=9, seed=7, width=1920, height=1080, log2_ctb=6, bit_depth=8, **bench.NATURAL)
aus, _ = ps.generate(ps.StreamParams(**kw))
out = {"what": "one 1080p intra picture of the encoder-like stream, one thread, ms from the call to the picture: `first` = first picture of a fresh decoder "
               "(its set-up included), `steady` = the same picture opening the second pass through that decoder; best of 5 decoders"}
for kind in (sys.argv[1:] or ["c", "sse", "null", "hip"]):
    if not ps.have(kind):
        continue
    first, steady = [], []

    def one_picture(d, pts):
        t = time.perf_counter()
        n = d.L.ohdec_decode(d.h, aus[0], len(aus[0]), pts)
        while n == 0:
            n = d.L.ohdec_flush(d.h)
            if n == 0:
                raise RuntimeError("the decoder gave no picture")
        if n < 0:
            raise RuntimeError(f"decode error {n}")
        return time.perf_counter() - t
    for _ in range(5):
        with ps.Decoder(kind, 1, 1) as d:
            first.append(one_picture(d, 1))
        with ps.Decoder(kind, 1, 1) as d:
            for i, au in enumerate(aus):
                d.L.ohdec_decode(d.h, au, len(au), i + 1)
            while d.L.ohdec_flush(d.h) > 0:
                pass
            steady.append(one_picture(d, 100))
    out[kind] = {"first": round(1e3 * min(first), 3), "steady": round(1e3 * min(steady), 3)}
print(json.dumps(out))
