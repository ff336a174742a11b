#!/usr/bin/env python3
"""Diagnose the drop-in test: which ops cover the mismatching samples (reference tables vs hooked tables)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import pyoracle as po
from openhevc_amd import lib as L
import stream_exec as X, synth_stream as S

ref = po.load("ref"); lib = L.load_library()
for rep in range(2):
  for (bd, W, H, intra_frac) in [(10, 192, 128, 0.15)]:
    rng = np.random.default_rng(77 + bd)
    dt = np.uint16 if bd > 8 else np.uint8
    dims = X.chroma_dims(W, H)
    refs = [[np.ascontiguousarray(rng.integers(0, 1 << bd, size=d).astype(dt)) for d in dims] for _ in range(2)]
    cur0 = [rng.integers(0, 1 << bd, size=d).astype(dt) for d in dims]
    ops, fops = S.gen_frame_ops(rng, W, H, bd, intra_frac=intra_frac, pcm_frac=0.04)
    for stage, ff in (("recon", []), ("all", fops)):
        enc = X.encode_driver_ops(ops, ff)
        want = [p.copy() for p in cur0]
        assert X.drive_tables(ref.lib, bd, W, H, want, refs, enc) == 0
        ctx = L.Ctx(0); slots = []
        for r in refs:
            s = ctx.pic_alloc(W, H, 1, bd); ctx.pic_upload(s, r); slots.append(s)
        cur_slot = ctx.pic_alloc(W, H, 1, bd); got = [p.copy() for p in cur0]; ctx.pic_upload(cur_slot, got)
        def reg(slot, planes):
            data = (C.c_void_p * 3)(*[p.ctypes.data for p in planes]); ls = (C.c_int * 3)(*[p.strides[0] for p in planes])
            L.check(lib.ohevc_tables_register_picture(ctx.h, slot, data, ls))
        for s, r in zip(slots, refs): reg(s, r)
        reg(cur_slot, got)
        L.check(lib.ohevc_tables_bind(ctx.h)); L.check(lib.ohevc_tables_begin_frame(ctx.h, cur_slot))
        geom = L.IntraGeom(W, H, 1, 6, 2, 1, 0, 0)
        hook = lambda f: C.cast(f, C.c_void_p).value
        rc = X.drive_tables(ref.lib, bd, W, H, got, refs, enc, hevcdsp_hook=hook(lib.ohevc_hevcdsp_init_hip), videodsp_hook=hook(lib.ohevc_videodsp_init_hip),
                            intra_hook=hook(lib.ohevc_tables_intra_pred), geom=C.addressof(geom))
        L.check(lib.ohevc_tables_end_frame(ctx.h, 1)); lib.ohevc_tables_bind(None); ctx.close()
        for c in range(3):
            bad = np.argwhere(got[c] != want[c])
            print(f"rep {rep} stage={stage} plane {c}: {len(bad)} mismatches", bad[:3].tolist())
            if len(bad) and stage == "recon":
                y, x = bad[0]; sh = 1 if c else 0
                hits = [o for o in ops if (o["t"] == "mc" and o["x0"] >> sh <= x < (o["x0"] + o["w"]) >> sh and o["y0"] >> sh <= y < (o["y0"] + o["h"]) >> sh) or
                        (o["t"] != "mc" and o["c_idx"] == c and o["x0"] >> sh <= x < (o["x0"] >> sh) + (1 << o["log2"]) and o["y0"] >> sh <= y < (o["y0"] >> sh) + (1 << o["log2"]))]
                print("   got", int(got[c][y, x]), "want", int(want[c][y, x]), [{k: v for k, v in h.items() if k not in ('coeffs', 'samples')} for h in hits][:4])
