#!/usr/bin/env python3
"""Diagnose ctx-vs-oracle mismatches stage by stage (reconstruction only / + deblock / + SAO) and attribute them."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import pyoracle as po
from openhevc_amd import lib as L
import stream_exec as X
import synth_stream as S

oracle = po.load("oracle")
for (bd, W, H, intra_frac) in [(8, 416, 240, 0.15), (10, 192, 136, 0.5)]:
    rng = np.random.default_rng(bd * 1000 + W)
    dt = np.uint16 if bd > 8 else np.uint8
    dims = X.chroma_dims(W, H)
    refs = [[rng.integers(0, 1 << bd, size=d).astype(dt) for d in dims] for _ in range(2)]
    cur0 = [rng.integers(0, 1 << bd, size=d).astype(dt) for d in dims]
    ops, fops = S.gen_frame_ops(rng, W, H, bd, n_refs=2, intra_frac=intra_frac)
    stages = {"recon": [], "recon+dbkV": [f for f in fops if f["t"] == "dbk" and f["vertical"]],
              "recon+dbk": [f for f in fops if f["t"] == "dbk"], "all": fops}
    for name, ff in stages.items():
        want = X.run_oracle(oracle, po, bd, W, H, [p.copy() for p in cur0], refs, ops, ff)
        ctx = L.Ctx(0)
        slots = []
        for r in refs:
            s = ctx.pic_alloc(W, H, 1, bd); ctx.pic_upload(s, r); slots.append(s)
        cur = ctx.pic_alloc(W, H, 1, bd); ctx.pic_upload(cur, cur0)
        ctx.frame_begin(cur); X.record_gpu(ctx, W, H, slots, ops, ff); ctx.frame_end()
        got = ctx.pic_download(cur, dims, dt); ctx.close()
        for c in range(3):
            bad = np.argwhere(got[c] != want[c])
            print(f"bd={bd} {W}x{H} stage={name} plane={c}: {len(bad)} mismatches", bad[:6].tolist())
            if len(bad) and name != "recon":
                for (y, x) in bad[:5]:
                    hits = [f for f in ff if f["c_idx"] == c and ((f["t"] == "sao" and f["x"] <= x < f["x"] + f["w"] and f["y"] <= y < f["y"] + f["h"]) or
                            (f["t"] == "dbk" and ((f["vertical"] and f["x"] - 4 <= x < f["x"] + 4 and f["y"] <= y < f["y"] + 8) or
                                                  (not f["vertical"] and f["y"] - 4 <= y < f["y"] + 4 and f["x"] <= x < f["x"] + 8))))]
                    print("   ", (int(y), int(x)), "got", int(got[c][y, x]), "want", int(want[c][y, x]), [{k: v for k, v in h.items() if k != 'offset_val'} for h in hits][:3])
            if len(bad) and name == "recon":
                for (y, x) in bad[:5]:
                    sh = 1 if c else 0
                    hits = [o for o in ops if (o["t"] == "mc" and o["x0"] >> sh <= x < (o["x0"] + o["w"]) >> sh and o["y0"] >> sh <= y < (o["y0"] + o["h"]) >> sh) or
                            (o["t"] != "mc" and o["c_idx"] == c and o["x0"] >> sh <= x < (o["x0"] >> sh) + (1 << o["log2"]) and o["y0"] >> sh <= y < (o["y0"] >> sh) + (1 << o["log2"]))]
                    print("   ", (int(y), int(x)), "got", int(got[c][y, x]), "want", int(want[c][y, x]), [{k: v for k, v in h.items() if k != 'coeffs'} for h in hits][:4])
