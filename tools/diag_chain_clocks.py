#!/usr/bin/env python3
"""Where a level of the intra chain kernel goes (include/ohevc_debug.h: ohevc_debug_intra_chain_clocks): decode the natural / flat 1080p stream
with one thread and print wavefront 0's shader clocks per level in its four phases.   python tools/diag_chain_clocks.py [natural]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pystream as ps  # noqa: E402

natural = "natural" in sys.argv[1:]
kw = dict(gop="random_access", nframes=9, seed=7, width=1920, height=1080, log2_ctb=6)
if natural:
    kw.update(init_qp=32, probs=dict(pred_mode=0.03, skip=0.55, merge_flag=0.7, split_cu=0.3, rqt_root_cbf=0.45, cbf_luma=0.5, cbf_chroma=0.25,
                                      split_transform=0.25, sig_coeff=0.35, last_x=0.5, last_y=0.5))
aus, _ = ps.generate(ps.StreamParams(**kw))
ps.decode_stream("hip", aus[:2], 1, 1)
prod = ps._product_lib()
for a in sys.argv[1:]:                       # name=value: ohevc_debug_set_<name>(value), e.g. chain_handover=2 (the wait for a level's stores of rounds 4-5)
    if "=" in a:
        getattr(prod, "ohevc_debug_set_" + a.split("=")[0])(int(a.split("=")[1]))
out = (C.c_ulonglong * 64)()
prod.ohevc_debug_intra_chain_clocks(1, None)
ps.decode_stream("hip", aus, 1, 1)
prod.ohevc_debug_intra_chain_clocks(1, out)
prod.ohevc_debug_intra_chain_clocks(0, None)
lv = max(1, out[4])
print(json.dumps(dict(stream="natural" if natural else "flat", switches=[a for a in sys.argv[1:] if "=" in a], levels=int(out[4]), clocks_per_level=dict(
    stores_and_barrier=round(out[0] / lv, 1), issue_loads_and_prefetches=round(out[1] / lv, 1), arithmetic_incl_sample_wait=round(out[2] / lv, 1),
    further_passes=round(out[3] / lv, 1), total=round(sum(out[:4]) / lv, 1), issue_after_sample_loads=round(out[5] / lv, 1),
    issue_after_residual_prefetch=round(out[6] / lv, 1), issue_after_level_record=round(out[7] / lv, 1)),
    per_size_class={f"{4 << k}x{4 << k}": dict(steps=int(out[16 + k]), arithmetic_clocks_per_step=round(out[8 + k] / max(1, out[16 + k]), 1),
                                              of_which_waiting_for_samples=round(out[12 + k] / max(1, out[16 + k]), 1)) for k in range(4)},
    levels_by_wavefronts={str(n) if n < 16 else ">=16": int(out[20 + n]) for n in range(17) if out[20 + n]}, note="shader clock (s_memtime): 100 MHz on gfx950 if constant, else core clock")))
