#!/usr/bin/env python3
"""A few synthetic streams through the reference's decoder with the HIP tables behind it, against the untouched decoder: the
shortest possible GPU visit for a new syntax knob.     python tools/check_streams_quick.py '<json list of StreamParams kwargs>'"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import pystream as ps

for kw in json.loads(sys.argv[1]):
    aus, gen = ps.generate(ps.StreamParams(**kw))
    ref = ps.decode_stream("c", aus)
    hip = ps.decode_stream("hip", aus)
    same = len(ref) == len(hip) and all(np.array_equal(x, y) for fa, fb in zip(ref, hip) for x, y in zip(fa, fb))
    print(json.dumps({"params": kw, "bit_exact": bool(same)}))
