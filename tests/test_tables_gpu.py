"""-m gpu: THE drop-in test.  The same miniature front-end (oracle/table_driver.c, compiled against the reference's real
headers) drives (a) the tables as the reference fills them, computing on host memory, and (b) the same tables after the
product's hooks ohevc_hevcdsp_init_hip / ohevc_videodsp_init_hip overrode them, recording into an ohevc_ctx and executing
on the GPU.  The two host frames must be identical after ohevc_tables_end_frame(download)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from oracle import pyoracle as po
from openhevc_amd import lib as L
import stream_exec as X

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import synth_stream as S  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd,W,H,intra_frac", [(8, 256, 136, 0.3), (10, 192, 128, 0.15)])
def test_reference_front_end_on_hooked_tables(ref, bd, W, H, intra_frac):
    rng = np.random.default_rng(77 + bd)
    dt = np.uint16 if bd > 8 else np.uint8
    dims = X.chroma_dims(W, H)
    refs = [[np.ascontiguousarray(rng.integers(0, 1 << bd, size=d).astype(dt)) for d in dims] for _ in range(2)]
    cur0 = [rng.integers(0, 1 << bd, size=d).astype(dt) for d in dims]
    ops, fops = S.gen_frame_ops(rng, W, H, bd, intra_frac=intra_frac, pcm_frac=0.04)
    assert any(o["t"] == "pcm" for o in ops)
    enc = X.encode_driver_ops(ops, fops)

    # (a) reference tables, host compute
    want = [p.copy() for p in cur0]
    assert X.drive_tables(ref.lib, bd, W, H, want, refs, enc) == 0

    # (b) hooked tables, GPU compute
    lib = L.load_library()
    ctx = L.Ctx(0)
    slots = []
    for r in refs:
        s = ctx.pic_alloc(W, H, 1, bd); ctx.pic_upload(s, r); slots.append(s)
    cur_slot = ctx.pic_alloc(W, H, 1, bd)
    got = [p.copy() for p in cur0]
    ctx.pic_upload(cur_slot, got)                     # the target's initial host content (pixels no job writes must survive)

    def reg(slot, planes):
        data = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
        ls = (C.c_int * 3)(*[p.strides[0] for p in planes])
        L.check(lib.ohevc_tables_register_picture(ctx.h, slot, data, ls))
    for s, r in zip(slots, refs):
        reg(s, r)
    reg(cur_slot, got)
    # the driver indexes references 0..n-1; picture-store slots were handed out in the same order
    assert slots == [0, 1]
    L.check(lib.ohevc_tables_bind(ctx.h))
    L.check(lib.ohevc_tables_begin_frame(ctx.h, cur_slot))
    geom = L.IntraGeom(W, H, 1, 6, 2, 1, 0, 0)
    hook = lambda f: C.cast(f, C.c_void_p).value
    rc = X.drive_tables(ref.lib, bd, W, H, got, refs, enc, hevcdsp_hook=hook(lib.ohevc_hevcdsp_init_hip),
                        videodsp_hook=hook(lib.ohevc_videodsp_init_hip), intra_hook=hook(lib.ohevc_tables_intra_pred),
                        geom=C.addressof(geom))
    assert rc == 0
    assert lib.ohevc_tables_status(ctx.h) == 0, lib.ohevc_last_error()
    # nothing was computed on the host: the frame is still untouched until end_frame downloads it
    assert all(np.array_equal(a, b) for a, b in zip(got, cur0))
    L.check(lib.ohevc_tables_end_frame(ctx.h, 1))
    st = ctx.stats()
    lib.ohevc_tables_bind(None)
    ctx.close()
    assert st["n_mc"] > 0 and st["n_tu"] > 0 and st["n_intra"] > 0 and st["n_dbk"] > 0 and st["n_sao"] > 0
    for c in range(3):
        bad = np.argwhere(got[c] != want[c])
        assert bad.size == 0, f"plane {c}: {len(bad)} mismatches, first {bad[:4].tolist()}; {st}"
