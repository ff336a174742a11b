"""-m gpu: a whole synthetic picture through the ctx layer (recorder + phase-ordered executor) vs the CPU oracle run
strictly in decode order.  Validates ordering rules (MC -> inter residual -> intra levels -> V edges -> H edges -> SAO)."""
import os
import sys

import numpy as np
import pytest

from oracle import pyoracle as po
from openhevc_amd import lib as L
import gpu_util as G
import stream_exec as X

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import synth_stream as S  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 1, 3, 2], ids=["launch_per_level", "levels_kernel", "ctb_tasks", "chosen_per_picture"])
def level_executor(request):
    """Every executor of the intra-coded blocks (include/ohevc_debug.h) must give the same pictures."""
    import ctypes
    if request.param == 1 and G.emulating():
        pytest.skip("the persistent level kernel's workgroups wait for each other; the emulator runs them one after another")
    lib = L.load_library()
    lib.ohevc_debug_set_level_launch.argtypes = [ctypes.c_int]
    prev = lib.ohevc_debug_set_level_launch(request.param)
    yield request.param
    lib.ohevc_debug_set_level_launch(prev)


@pytest.mark.parametrize("bd,W,H,intra_frac", [(8, 416, 240, 0.15), (10, 192, 136, 0.5), (8, 128, 128, 1.0), (14, 192, 136, 0.3)])
def test_synthetic_picture_matches_decode_order_oracle(oracle, level_executor, bd, W, H, intra_frac):
    rng = np.random.default_rng(bd * 1000 + W)
    dt = G.pixdt(bd)
    dims = X.chroma_dims(W, H)
    refs = [[rng.integers(0, 1 << bd, size=d).astype(dt) for d in dims] for _ in range(2)]
    cur0 = [rng.integers(0, 1 << bd, size=d).astype(dt) for d in dims]
    ops, fops = S.gen_frame_ops(rng, W, H, bd, n_refs=2, intra_frac=intra_frac)
    want = X.run_oracle(oracle, po, bd, W, H, [p.copy() for p in cur0], refs, ops, fops)

    ctx = L.Ctx(0)
    ref_slots = []
    for r in refs:
        s = ctx.pic_alloc(W, H, 1, bd); ctx.pic_upload(s, r); ref_slots.append(s)
    cur = ctx.pic_alloc(W, H, 1, bd)
    ctx.pic_upload(cur, cur0)
    ctx.frame_begin(cur)
    X.record_gpu(ctx, W, H, ref_slots, ops, fops)
    ctx.frame_end()
    got = ctx.pic_download(cur, dims, dt)
    st = ctx.stats()
    ctx.close()
    assert st["n_mc"] + st["n_intra"] > 0 and st["launches"] > 0
    for c in range(3):
        bad = np.argwhere(got[c] != want[c])
        assert bad.size == 0, f"plane {c}: {len(bad)} mismatches, first {bad[:4].tolist()}; stats {st}"


def test_reconstruct_in_several_flushes(oracle):
    """Calling ohevc_frame_reconstruct after every CTU row (the reference's natural flush point) gives the same picture."""
    bd, W, H = 8, 256, 192
    rng = np.random.default_rng(99)
    dims = X.chroma_dims(W, H)
    refs = [[rng.integers(0, 256, size=d).astype(np.uint8) for d in dims] for _ in range(2)]
    cur0 = [rng.integers(0, 256, size=d).astype(np.uint8) for d in dims]
    ops, fops = S.gen_frame_ops(rng, W, H, bd, intra_frac=0.4)
    want = X.run_oracle(oracle, po, bd, W, H, [p.copy() for p in cur0], refs, ops, [])
    ctx = L.Ctx(0)
    slots = []
    for r in refs:
        s = ctx.pic_alloc(W, H, 1, bd); ctx.pic_upload(s, r); slots.append(s)
    cur = ctx.pic_alloc(W, H, 1, bd); ctx.pic_upload(cur, cur0)
    ctx.frame_begin(cur)
    row = 0
    chunk = []
    for op in ops:
        r = op["y0"] // 64
        if r != row:
            X.record_gpu(ctx, W, H, slots, chunk, []); ctx.frame_reconstruct(); chunk = []; row = r
        chunk.append(op)
    X.record_gpu(ctx, W, H, slots, chunk, [])
    ctx.frame_end()
    got = ctx.pic_download(cur, dims, np.uint8)
    ctx.close()
    for c in range(3):
        assert np.array_equal(got[c], want[c]), c


@pytest.mark.parametrize("bd,cfi", [(8, 1), (10, 2), (14, 3)])
def test_picture_export_import_round_trip(bd, cfi):
    """ohevc_pic_export / ohevc_pic_import (frame-parallel decoding over GPUs, include/ohevc_ctx.h): a picture leaves one store slot
    through a device buffer laid out like the store's plane (stride x height) and enters another slot bit for bit."""
    rng = np.random.default_rng(4100 + bd)
    W, H = 200, 72
    hs, vs = int(cfi in (1, 2)), int(cfi == 1)
    dt = G.pixdt(bd)
    planes = [rng.integers(0, 1 << bd, size=s).astype(dt) for s in ((H, W), (H >> vs, W >> hs), (H >> vs, W >> hs))]
    ctx = L.Ctx()
    a, b = ctx.pic_alloc(W, H, cfi, bd), ctx.pic_alloc(W, H, cfi, bd)
    ctx.pic_upload(a, planes)
    ctx.pic_upload(b, [np.zeros_like(p) for p in planes])
    geo = ctx.pic_planes(a)
    for c in range(3):
        n = int(geo[c].stride) * int(geo[c].height)
        buf = G.zeros_dev(n, np.uint8)
        ctx.pic_export(a, c, buf.data_ptr(), n)
        ctx.pic_import(b, c, buf.data_ptr(), n)
        with pytest.raises(RuntimeError):
            ctx.pic_export(a, c, buf.data_ptr(), n - 1)         # the size is part of the contract
    got = ctx.pic_download(b, [p.shape for p in planes], dt)
    for c in range(3):
        assert np.array_equal(got[c], planes[c]), f"plane {c}"
    ctx.close()


def test_device_pictures_come_in_batches_per_size():
    """Device pictures are allocated, zeroed and waited for in batches of 4, 8, 16, 32 per piece size (ctx.hip: PicStore::spare): a decoder's
    frame-buffer pool grows one picture at a time, in the serial prologue of its pictures.  Two sizes taking turns - the two layers of an SHVC
    stream share a store - must not start a new batch with every picture; pictures of a batch are distinct memory, zeroed, and stay usable
    after neighbours were released."""
    import ctypes
    lib = L.load_library()
    lib.ohevc_debug_picture_batches.argtypes = [ctypes.c_void_p]
    if os.environ.get("OHEVC_PICTURE_BATCH") == "0":
        pytest.skip("batches are switched off in this run")
    ctx = L.Ctx(0)
    assert lib.ohevc_debug_picture_batches(ctx.h) == 0
    sizes = [(192, 128), (384, 256)]
    slots = []
    for i in range(40):
        w, h = sizes[i & 1]
        slots.append((ctx.pic_alloc(w, h, 1, 8), w, h))
    # 20 pictures of each size: batches of 4 + 8 + 16 per size
    assert lib.ohevc_debug_picture_batches(ctx.h) == 6
    rng = np.random.default_rng(5)
    kept = {}
    for k, (slot, w, h) in enumerate(slots):
        shapes = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
        got = ctx.pic_download(slot, shapes, np.uint8)
        assert all(not pl.any() for pl in got), "a fresh device picture is zeroed (like the reference's frame pool)"
        planes = [rng.integers(0, 256, size=sh).astype(np.uint8) for sh in shapes]
        ctx.pic_upload(slot, planes)
        kept[slot] = (planes, shapes)
    for slot, w, h in slots[::3]:
        ctx.pic_release(slot)
        kept.pop(slot)
    for slot, (planes, shapes) in kept.items():             # every picture still holds what was written to it: the pieces do not overlap
        got = ctx.pic_download(slot, shapes, np.uint8)
        assert all(np.array_equal(a, b) for a, b in zip(got, planes))
    ctx.close()


def test_released_pieces_are_handed_out_again_zeroed():
    """release + alloc cycles (a decoder's pool changing geometry, an enhancement layer reopened on a live base store) must not grow device
    memory: the piece of a released picture goes back to the store and is zeroed when it is handed out again - before any new batch is made."""
    import ctypes
    lib = L.load_library()
    lib.ohevc_debug_picture_batches.argtypes = [ctypes.c_void_p]
    if os.environ.get("OHEVC_PICTURE_BATCH") == "0":
        pytest.skip("batches are switched off in this run")
    ctx = L.Ctx(0)
    w, h = 320, 192
    shapes = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
    rng = np.random.default_rng(9)
    live = [ctx.pic_alloc(w, h, 1, 8) for _ in range(4)]    # the first batch of 4
    assert lib.ohevc_debug_picture_batches(ctx.h) == 1
    keep_planes = [rng.integers(1, 256, size=sh).astype(np.uint8) for sh in shapes]
    ctx.pic_upload(live[0], keep_planes)
    for cycle in range(100):                                 # 100 pictures come and go through the other three pieces
        slot = live.pop()
        ctx.pic_upload(slot, [np.full(sh, 1 + cycle % 255, np.uint8) for sh in shapes])
        ctx.pic_release(slot)
        slot = ctx.pic_alloc(w, h, 1, 8)
        got = ctx.pic_download(slot, shapes, np.uint8)
        assert all(not pl.any() for pl in got), "a recycled piece is zeroed like a fresh one"
        live.append(slot)
    assert lib.ohevc_debug_picture_batches(ctx.h) == 1, "release + alloc cycles made new batches: released pieces are not reused"
    got = ctx.pic_download(live[0], shapes, np.uint8)
    assert all(np.array_equal(a, b) for a, b in zip(got, keep_planes)), "a neighbour's piece was touched by the recycling"
    ctx.close()


def test_cached_stream_handle_is_ordered_behind_a_long_chain_picture(oracle):
    """ohevc_ctx_stream is one handle for the context's life (include/ohevc_ctx.h).  A picture whose dependency levels reach
    ohevc_debug_set_long_chain_levels is issued on the context's second, internal stream; its frame end joins the public stream again, so a
    caller that cached the handle and synchronises on IT (not on ohevc_ctx_sync) reads finished pixels.  The pictures are ADOPTED
    planes (ohevc_pic_adopt) with the owner's pitch - half the store's for the chroma planes here: the deblocked copy SAO reads has the store's
    pitch (a plane-sized linear copy between the two was wrong until round 6).  On the emulator (streams are synchronous) this is the adopted-pitch check alone."""
    import ctypes
    emu = G.emulating()
    lib = L.load_library()
    lib.ohevc_debug_set_long_chain_levels.argtypes = [ctypes.c_int]
    lib.ohevc_ctx_stream.restype = ctypes.c_void_p
    bd, W, H = 8, 256, 256
    rng = np.random.default_rng(77)
    dims = X.chroma_dims(W, H)
    cur0 = [rng.integers(0, 256, size=d).astype(np.uint8) for d in dims]
    refs = [[rng.integers(0, 256, size=d).astype(np.uint8) for d in dims] for _ in range(2)]
    ops, fops = S.gen_frame_ops(rng, W, H, bd, n_refs=2, intra_frac=0.9)
    want = X.run_oracle(oracle, po, bd, W, H, [p.copy() for p in cur0], refs, ops, fops)
    lib.ohevc_debug_set_long_chain_levels(1)                   # every picture with a level at all is a long chain
    try:
        ctx = L.Ctx(0)
        handle = lib.ohevc_ctx_stream(ctx.h)
        if not emu:
            import torch
            ext = torch.cuda.ExternalStream(handle)
        ref_slots = []
        for r in refs:
            sl = ctx.pic_alloc(W, H, 1, bd); ctx.pic_upload(sl, r); ref_slots.append(sl)
        for rep in range(2 if emu else 6):                     # the handle survives pictures on either stream
            planes = [G.to_dev(p.copy()) for p in cur0]
            G.sync()
            cur = ctx.pic_adopt(planes, W, H, 1, bd)
            ctx.frame_begin(cur)
            X.record_gpu(ctx, W, H, ref_slots, ops, fops)
            ctx.frame_end()
            assert lib.ohevc_ctx_stream(ctx.h) == handle
            if not emu:
                ext.synchronize()                              # the CACHED handle, nothing else
            got = [G.to_host(t, np.uint8) for t in planes]
            for c in range(3):
                assert np.array_equal(got[c], want[c]), f"pass {rep}, plane {c}: read unfinished (or wrongly copied) pixels through the cached stream handle"
            ctx.pic_release(cur)
        assert ctx.stats()["launches"] > 0
        ctx.close()
    finally:
        lib.ohevc_debug_set_long_chain_levels(96)
