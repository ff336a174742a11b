"""-m gpu: ohevc_dev_expand_coeffs on its own - the device half of the compact coefficient stream (include/ohevc_hip.h: ohevc_expand_rec) - and
ohevc_rec_tu_limited end to end, on the corners of the col_limit argument the reference hands its inverse transforms (hevcdsp.h:53).

The oracle's idct (oracle/hevc_oracle.c, the restatement of hevcdsp_template.c:210-316) reads column c of the block only for c < col_limit and row r
only for r < min(col_limit + 4, N) (:271-291): a compact stream that keeps exactly what it reads - the rectangle, or the non-zero 4x4 groups inside
it - must reconstruct the same pictures as the dense block."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po
from openhevc_amd import lib as L
import gpu_util as G

pytestmark = pytest.mark.gpu

REC = np.dtype([("src", "<u4"), ("dst", "<u4"), ("dims", "<u4"), ("kind", "<u4")])


def _expand(compact, recs, dense_len):
    lib = L.load_library()
    d_c, d_r = G.to_dev(compact), G.to_dev(recs)
    d_out = G.to_dev(np.full(dense_len, 0x5a5a, np.int16))          # the kernel must write every element of every block, zeros included
    L.check(lib.ohevc_dev_expand_coeffs(C.c_void_p(d_c.data_ptr()), C.c_void_p(d_r.data_ptr()), C.c_int(len(recs)), C.c_void_p(d_out.data_ptr()), C.c_void_p(G.stream())))
    G.sync()
    return G.to_host(d_out, np.int16)


def _pack(blocks):
    """blocks: [(log2, dense N x N int16 block, form, cols, rows)], form in 'whole' / 'rect' / 'groups' -> (compact, recs, dense expected)"""
    compact, recs, dense = [], [], []
    src = dst = 0
    for log2, blk, form, cols, rows in blocks:
        n = 1 << log2
        if form == "whole":
            compact.append(blk.ravel()); recs.append((src, dst, n * n, 0)); src += n * n
            want = blk
        elif form == "rect":
            compact.append(blk[:rows, :cols].ravel()); recs.append((src, dst, cols | rows << 8, log2)); src += cols * rows
            want = np.zeros_like(blk); want[:rows, :cols] = blk[:rows, :cols]
        else:
            want = np.zeros_like(blk)
            parts = 2 if (log2 == 5 and rows > 16) else 1
            for part in range(parts):
                gy0, gy1 = part * 4, min(rows // 4, part * 4 + 4 if log2 == 5 else n // 4)
                mask, at = 0, src
                for gy in range(gy0, gy1):
                    for gx in range(cols // 4):
                        g4 = blk[4 * gy:4 * gy + 4, 4 * gx:4 * gx + 4]
                        if not g4.any():
                            continue
                        compact.append(g4.ravel()); src += 16
                        mask |= 1 << ((gy - gy0) * (n // 4) + gx)
                        want[4 * gy:4 * gy + 4, 4 * gx:4 * gx + 4] = g4
                code = 0 if log2 != 5 else (part if parts == 2 else 2)
                recs.append((at, dst + part * 512, mask, 0x100 | log2 | code << 9))
        dense.append(want.ravel()); dst += n * n
    return (np.concatenate(compact).astype(np.int16) if compact else np.zeros(16, np.int16)), np.array(recs, REC), np.concatenate(dense)


def _block(rng, log2, cols, rows, density):
    n = 1 << log2
    b = np.zeros((n, n), np.int16)
    live = rng.random((rows, cols)) < density
    b[:rows, :cols] = np.where(live, rng.integers(-32768, 32768, (rows, cols)), 0)
    return b


def test_expand_kernel_every_form_and_corner():
    rng = np.random.default_rng(6)
    blocks = []
    for log2 in (2, 3, 4, 5):
        n = 1 << log2
        blocks.append((log2, _block(rng, log2, n, n, 1.0), "whole", n, n))
        if log2 == 2:
            continue
        # col_limit corners: the smallest rectangle, full width with few rows, rectangles that end in the middle of a group row of a 32x32 half,
        # the whole block as a rectangle, and (groups) blocks whose only non-zero group is the last one / none at all
        for cols, rows in {(4, 4), (4, 8), (n, 4), (4, n), (n // 2, n // 2 + 4 if n // 2 + 4 <= n else n), (n, n), (n - 4, n), (n, n - 4), (8, 20 if n == 32 else 8), (n, 16 if n == 32 else n)}:
            if cols > n or rows > n:
                continue
            for density in (1.0, 0.3, 0.02, 0.0):
                blk = _block(rng, log2, cols, rows, density)
                blocks.append((log2, blk, "rect", cols, rows))
                blocks.append((log2, blk, "groups", cols, rows))
        last = np.zeros((n, n), np.int16); last[n - 1, n - 1] = -7
        blocks.append((log2, last, "groups", n, n))
        first = np.zeros((n, n), np.int16); first[0, 0] = 9
        blocks.append((log2, first, "groups", 4, 4))
    order = rng.permutation(len(blocks))
    blocks = [blocks[k] for k in order]
    compact, recs, want = _pack(blocks)
    got = _expand(compact, recs, len(want))
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, f"{bad.size} elements differ, first at {bad[:5].tolist()}"


@pytest.mark.parametrize("compact_mode", [2, 1, 0], ids=["groups", "rectangles", "dense"])
@pytest.mark.parametrize("bd", [8, 10])
def test_rec_tu_limited_matches_oracle_col_limit(oracle, compact_mode, bd):
    """ohevc_rec_tu_limited through the ctx layer (recorder -> upload -> expand -> TU kernels) against the oracle's idct WITH the reference's col_limit
    read pattern: the bound is exact where the coefficients outside it are zero (the caller's promise), loose bounds are fine too."""
    lib = L.load_library()
    lib.ohevc_debug_set_compact_coeffs.argtypes = [C.c_int]
    lib.ohevc_debug_set_compact_coeffs(compact_mode)
    try:
        rng = np.random.default_rng(60 + bd)
        W, H = 256, 192
        dt = G.pixdt(bd)
        plane0 = rng.integers(0, 1 << bd, (H, W)).astype(dt)
        want = plane0.copy()
        ctx = L.Ctx(0)
        slot = ctx.pic_alloc(W, H, 1, bd)
        ctx.pic_upload(slot, [plane0, np.zeros((H // 2, W // 2), dt), np.zeros((H // 2, W // 2), dt)])
        ctx.frame_begin(slot)
        lib.ohevc_rec_tu_limited.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        y = 0
        for log2 in (3, 4, 5):
            n = 1 << log2
            x = 0
            for col_limit in [4, 8, n // 2, n - 4, n] + [4 * int(rng.integers(1, n // 4 + 1)) for _ in range(3)]:      # (the reference's bounds are multiples of 4, hevc_cabac.c:1923-1934)
                cl = min(n, (col_limit + 3) & ~3)
                cols, rows = min(cl, n), min(cl + 4, n)
                blk = _block(rng, log2, cols, rows, float(rng.choice([1.0, 0.2, 0.03])))
                blk = np.clip(blk, -2048, 2047).astype(np.int16)
                # what the reference's first pass READS: column i down to row limit2, which shrinks by 4 after columns 4, 8, ... while it is below N
                # (hevcdsp_template.c:283-288) - coefficients outside that staircase cannot be non-zero in a stream (the scan is diagonal)
                limit2 = min(cl + 4, n)
                for i in range(n):
                    blk[limit2:, i] = 0
                    if limit2 < n and i % 4 == 0 and i:
                        limit2 -= 4
                L.check(lib.ohevc_rec_tu_limited(ctx.h, 0, x, y, log2, L.TU_IDCT, blk.ctypes.data_as(C.c_void_p), 0, col_limit, min(col_limit + 4, n)))
                want = oracle.tu_batch(bd, po.TU_IDCT, log2, blk.reshape(1, n, n), want, np.array([[x, y]], np.int32), col_limit=cl)
                x += n
            y += n
        ctx.frame_end()
        got = ctx.pic_download(slot, [(H, W), (H // 2, W // 2), (H // 2, W // 2)], dt)[0]
        ctx.close()
        assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} samples differ"
    finally:
        lib.ohevc_debug_set_compact_coeffs(2)
