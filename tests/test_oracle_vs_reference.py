"""Pin the CPU restatement (oracle/hevc_oracle.c) against the reference's own compiled C
(oracle/_ref/libhevcref.so = /root/reference/libavcodec/hevcdsp.c + hevcpred.c, unmodified).

The reference ships no tests or golden vectors, so this differential test plus the
fixtures in tests/golden/ (generated from the same reference build) are what pins the oracle.
"""
import numpy as np
import pytest

from oracle import pyoracle as po

BDS = [8, 10, 12, 14]


def pixdt(bd):
    return np.uint16 if bd > 8 else np.uint8


def rand_plane(rng, bd, h, w):
    return rng.integers(0, 1 << bd, size=(h, w)).astype(pixdt(bd))


# ------------------------------------------------------------------ transforms
@pytest.mark.parametrize("bd", BDS)
@pytest.mark.parametrize("log2", [2, 3, 4, 5])
def test_idct_dense(oracle, ref, bd, log2):
    rng = np.random.default_rng(100 + bd * 10 + log2)
    n = 1 << log2
    for amp in (1 << 15, 1024, 64):
        for _ in range(6):
            c = rng.integers(-amp, amp, size=(n, n)).astype(np.int16)
            assert np.array_equal(oracle.tu_residual(bd, po.TU_IDCT, log2, c), ref.tu_residual(bd, po.TU_IDCT, log2, c))
    # extreme values exercise both clip_int16 stages
    for v in (-32768, 32767):
        c = np.full((n, n), v, np.int16)
        assert np.array_equal(oracle.tu_residual(bd, po.TU_IDCT, log2, c), ref.tu_residual(bd, po.TU_IDCT, log2, c))


@pytest.mark.parametrize("bd", [8, 10, 14])
@pytest.mark.parametrize("log2", [2, 3, 4, 5])
def test_idct_col_limit_semantics(oracle, ref, bd, log2):
    """The partial butterflies skip inputs beyond col_limit (hevcdsp_template.c:264-301); the restatement
    must agree even on (contract-violating) dense input, and a full transform must agree on legal input."""
    rng = np.random.default_rng(7 + log2)
    n = 1 << log2
    for col_limit in sorted({4, 8, 12, 24, n} & set(range(4, n + 1)) | {n}):
        c = rng.integers(-2000, 2000, size=(n, n)).astype(np.int16)
        a = oracle.tu_residual(bd, po.TU_IDCT, log2, c, col_limit)
        b = ref.tu_residual(bd, po.TU_IDCT, log2, c, col_limit)
        assert np.array_equal(a, b), (col_limit,)
        # legal input: nonzeros only where x + y <= col_limit - 4 (hevc_cabac.c:1923-1934)
        yy, xx = np.mgrid[0:n, 0:n]
        legal = np.where(xx + yy <= col_limit - 4, c, 0).astype(np.int16)
        full = oracle.tu_residual(bd, po.TU_IDCT, log2, legal, n)
        assert np.array_equal(full, ref.tu_residual(bd, po.TU_IDCT, log2, legal, col_limit))


@pytest.mark.parametrize("bd", BDS)
def test_other_residual_kinds(oracle, ref, bd):
    rng = np.random.default_rng(5 + bd)
    for log2 in (2, 3, 4, 5):
        n = 1 << log2
        kinds = [po.TU_DC, po.TU_SKIP, po.TU_SKIP_RDPCM_H, po.TU_SKIP_RDPCM_V, po.TU_BYPASS,
                 po.TU_BYPASS_RDPCM_H, po.TU_BYPASS_RDPCM_V] + ([po.TU_DST4] if log2 == 2 else [])
        for kind in kinds:
            for amp in (1 << 15, 300):
                c = rng.integers(-amp, amp, size=(n, n)).astype(np.int16)
                assert np.array_equal(oracle.tu_residual(bd, kind, log2, c), ref.tu_residual(bd, kind, log2, c)), (kind, log2)


@pytest.mark.parametrize("bd", BDS)
@pytest.mark.parametrize("log2", [2, 3, 4, 5])
def test_tu_batch_add(oracle, ref, bd, log2):
    rng = np.random.default_rng(11 + bd + log2)
    n = 1 << log2
    nblk = 24
    plane = rand_plane(rng, bd, 4 * n, 6 * n + 8)
    xy = np.array([[(i % 6) * n + 4, (i // 6) * n] for i in range(nblk)], np.int32)
    c = rng.integers(-1024, 1024, size=(nblk, n, n)).astype(np.int16)
    a = oracle.tu_batch(bd, po.TU_IDCT, log2, c, plane.copy(), xy)
    b = ref.tu_batch(bd, po.TU_IDCT, log2, c, plane.copy(), xy)
    assert np.array_equal(a, b)
    assert np.array_equal(oracle.tu_batch(bd, po.TU_IDCT, log2, c, plane.copy(), xy, threads=3), b)
    assert np.array_equal(ref.tu_batch(bd, po.TU_IDCT, log2, c, plane.copy(), xy, threads=3), b)


# ------------------------------------------------------------------ motion compensation
WIDTHS_LUMA = [4, 8, 12, 16, 24, 32, 48, 64]
WIDTHS_CHROMA = [2, 4, 6, 8, 12, 16, 24, 32]


@pytest.mark.parametrize("bd", BDS)
@pytest.mark.parametrize("luma", [1, 0])
def test_mc_all_variants(oracle, ref, bd, luma):
    rng = np.random.default_rng(21 + bd + luma)
    refp = rand_plane(rng, bd, 96, 112)
    fr = 4 if luma else 8
    for w in (WIDTHS_LUMA if luma else WIDTHS_CHROMA):
        for (mx, my) in [(0, 0), (1, 0), (0, 2), (3, 1), (fr - 1, fr - 1)]:
            h = int(rng.choice([4, 8, 16, 24])) if w > 2 else 4
            sx, sy = int(rng.integers(4, 20)), int(rng.integers(4, 20))
            src2 = rng.integers(-8000, 16000, size=(h, 64)).astype(np.int16)
            kw = dict(denom=int(rng.integers(0, 8)), wx0=int(rng.integers(-128, 128)), wx1=int(rng.integers(-128, 128)),
                      ox0=int(rng.integers(-128, 128)), ox1=int(rng.integers(-128, 128)))
            for variant in (po.MC_PUT, po.MC_UNI, po.MC_UNI_W, po.MC_BI, po.MC_BI_W):
                a = oracle.mc(bd, luma, variant, refp, sx, sy, w, h, mx, my, src2=src2, **kw)
                b = ref.mc(bd, luma, variant, refp, sx, sy, w, h, mx, my, src2=src2, **kw)
                assert np.array_equal(a, b), (w, h, mx, my, variant, kw)


@pytest.mark.parametrize("bd", [b for b in BDS if b > 8])
@pytest.mark.parametrize("luma", [1, 0])
def test_mc_samples_above_the_bit_depth(oracle, ref, bd, luma):
    """Above 8 bit the reference's constrained intra prediction leaves samples of up to 0x8080 in its pictures (its byte-wise
    memset, hevcpred_template.c:117-141) and later pictures predict from them: the full-sample uni case copies them through,
    the int16 intermediates wrap.  What the reference computes there is what the path has to compute."""
    rng = np.random.default_rng(521 + bd + luma)
    refp = rand_plane(rng, bd, 96, 112)
    wild = rng.random(refp.shape) < 0.08
    refp[wild] = rng.integers(1 << bd, 0x8081, size=int(wild.sum())).astype(refp.dtype)
    refp[40:44, 40:60] = 0x8080
    fr = 4 if luma else 8
    for w in (WIDTHS_LUMA if luma else WIDTHS_CHROMA)[:5]:
        for (mx, my) in [(0, 0), (1, 0), (0, 2), (3, 1), (fr - 1, fr - 1)]:
            h = int(rng.choice([4, 8, 16])) if w > 2 else 4
            sx, sy = int(rng.integers(20, 44)), int(rng.integers(20, 40))
            kw = dict(denom=int(rng.integers(0, 8)), wx0=int(rng.integers(-128, 128)), wx1=int(rng.integers(-128, 128)),
                      ox0=int(rng.integers(-128, 128)), ox1=int(rng.integers(-128, 128)))
            src2 = oracle.mc(bd, luma, po.MC_PUT, refp, sx + 3, sy + 5, w, h, mx, my)
            assert np.array_equal(src2, ref.mc(bd, luma, po.MC_PUT, refp, sx + 3, sy + 5, w, h, mx, my))
            s2 = np.zeros((h, 64), np.int16); s2[:, :w] = src2
            for variant in (po.MC_UNI, po.MC_UNI_W, po.MC_BI, po.MC_BI_W):
                a = oracle.mc(bd, luma, variant, refp, sx, sy, w, h, mx, my, src2=s2, **kw)
                b = ref.mc(bd, luma, variant, refp, sx, sy, w, h, mx, my, src2=s2, **kw)
                assert np.array_equal(a, b), (w, h, mx, my, variant, kw)


# ------------------------------------------------------------------ deblocking
@pytest.mark.parametrize("bd", BDS)
def test_deblock(oracle, ref, bd):
    rng = np.random.default_rng(31 + bd)
    for it in range(300):
        # smooth-ish content so that all three decisions (off / normal / strong) occur
        base = rng.integers(0, 1 << bd)
        amp = int(rng.choice([1, 2, 4, 16, 64])) << (bd - 8)
        plane = np.clip(base + rng.integers(-amp, amp + 1, size=(24, 24)), 0, (1 << bd) - 1).astype(pixdt(bd))
        if it % 3 == 0:
            plane[:, 12:] = np.clip(plane[:, 12:].astype(int) + (int(rng.integers(-12, 13)) << (bd - 8)), 0, (1 << bd) - 1)
            plane[12:, :] = np.clip(plane[12:, :].astype(int) + (int(rng.integers(-12, 13)) << (bd - 8)), 0, (1 << bd) - 1)
        beta = int(rng.integers(0, 65)); tc = [int(rng.integers(0, 25)), int(rng.integers(0, 25))]
        no_p = [int(rng.random() < 0.15), int(rng.random() < 0.15)]; no_q = [int(rng.random() < 0.15), int(rng.random() < 0.15)]
        for vert in (0, 1):
            x, y = (12, 8) if vert else (8, 12)
            a, b = plane.copy(), plane.copy()
            oracle.deblock_luma(bd, vert, a, x, y, beta, tc, no_p, no_q)
            ref.deblock_luma(bd, vert, b, x, y, beta, tc, no_p, no_q)
            assert np.array_equal(a, b), (it, vert, beta, tc)
            a, b = plane.copy(), plane.copy()
            oracle.deblock_chroma(bd, vert, a, x, y, tc, no_p, no_q)
            ref.deblock_chroma(bd, vert, b, x, y, tc, no_p, no_q)
            assert np.array_equal(a, b), (it, vert, tc)


# ------------------------------------------------------------------ SAO
@pytest.mark.parametrize("bd", BDS)
def test_sao(oracle, ref, bd):
    rng = np.random.default_rng(41 + bd)
    for it in range(120):
        w, h = int(rng.choice([8, 16, 24, 32, 64])), int(rng.choice([8, 16, 32, 64]))
        src = rand_plane(rng, bd, h + 2, w + 2)
        if it % 2:
            src = (src >> (bd - 3)).astype(src.dtype) + (1 << (bd - 1))     # flat content: many equal neighbours
        ov = np.concatenate([[0], rng.integers(-31, 32, size=4) << (bd - 8 if bd <= 10 else 2)]).astype(np.int16)
        a = np.zeros_like(src); b = np.zeros_like(src)
        bp = int(rng.integers(0, 32))
        oracle.sao_band(bd, a, src, 1, 1, w, h, ov, bp); ref.sao_band(bd, b, src, 1, 1, w, h, ov, bp)
        assert np.array_equal(a, b)
        for eo in range(4):
            borders = [int(rng.random() < 0.3) for _ in range(4)]
            ve = [int(rng.random() < 0.3) for _ in range(2)]; he = [int(rng.random() < 0.3) for _ in range(2)]
            de = [int(rng.random() < 0.3) for _ in range(4)]
            for restore in (0, 1):
                a = np.zeros_like(src); b = np.zeros_like(src)
                oracle.sao_edge(bd, restore, a, src, 1, 1, w, h, ov, eo, borders, ve, he, de)
                ref.sao_edge(bd, restore, b, src, 1, 1, w, h, ov, eo, borders, ve, he, de)
                assert np.array_equal(a, b), (it, eo, restore, borders, ve, he, de)


# ------------------------------------------------------------------ intra
@pytest.mark.parametrize("bd", BDS)
@pytest.mark.parametrize("log2", [2, 3, 4, 5])
def test_pred_modes(oracle, ref, bd, log2):
    rng = np.random.default_rng(51 + bd + log2)
    n = 1 << log2
    for mode in range(35):
        for c_idx in (0, 1):
            top = rng.integers(0, 1 << bd, size=2 * n + 1 + 8)
            left = rng.integers(0, 1 << bd, size=2 * n + 1 + 8)
            a = oracle.pred(bd, log2, mode, top, left, c_idx); b = ref.pred(bd, log2, mode, top, left, c_idx)
            assert np.array_equal(a, b), (mode, c_idx)


@pytest.mark.parametrize("bd", [8, 10, 14])
def test_intra_pred_full(oracle, ref, bd):
    rng = np.random.default_rng(61 + bd)
    W, H = 136, 72                      # not CTB aligned: exercises picture-edge clipping of the 2N neighbours
    for it in range(400):
        log2 = int(rng.integers(2, 6)); n = 1 << log2
        c_idx = int(rng.integers(0, 3))
        cfi = int(rng.choice([1, 1, 1, 3]))
        sh = 1 if (c_idx and cfi == 1) else 0
        nl = n << sh
        x0 = int(rng.integers(0, (W - nl) // nl + 1)) * nl; y0 = int(rng.integers(0, (H - nl) // nl + 1)) * nl
        if it % 5 == 0:
            x0, y0 = (W - nl) // nl * nl, (H - nl) // nl * nl
        mode = int(rng.integers(0, 35))
        cands = [int(rng.random() < 0.7) for _ in range(5)]
        if x0 == 0: cands[0] = cands[1] = cands[2] = 0
        if y0 == 0: cands[2] = cands[3] = cands[4] = 0
        if x0 + nl >= W: cands[4] = 0
        if y0 + nl >= H: cands[0] = 0
        if it % 7 == 0:                  # smooth neighbourhood: triggers the strong 32x32 filter
            planes = [np.full((H + 8, W + 8), int(rng.integers(0, 1 << bd)), pixdt(bd)) + (np.arange(W + 8) // 16).astype(pixdt(bd)) for _ in range(3)]
            planes = [np.ascontiguousarray(p) for p in planes]
        else:
            planes = [rand_plane(rng, bd, H + 8, W + 8) for _ in range(3)]
        strong = int(rng.random() < 0.7); dis = int(rng.random() < 0.1)
        pa = [p.copy() for p in planes]; pb = [p.copy() for p in planes]
        kw = dict(chroma_format_idc=cfi, strong=strong, smoothing_disabled=dis,
                  log2_ctb_size=int(rng.choice([4, 5, 6])), log2_min_tb_size=2)
        oracle.intra_pred(bd, pa, W, H, x0, y0, log2, c_idx, mode, cands, **kw)
        ref.intra_pred(bd, pb, W, H, x0, y0, log2, c_idx, mode, cands, **kw)
        for i in range(3):
            assert np.array_equal(pa[i], pb[i]), (it, log2, c_idx, mode, cands, x0, y0, kw)


@pytest.mark.parametrize("bd", [8, 10, 14])
def test_intra_pred_constrained(oracle, ref, bd):
    """constrained_intra_pred_flag = 1: availability re-derivation + substitution walk (hevcpred_template.c:116-163,185-249).
    Positions are kept off the top picture row when the block has left neighbours: there the reference itself reads
    tab_mvf out of bounds (IS_INTRA(-1,-1) with y0 == 0), so its result is undefined."""
    rng = np.random.default_rng(71 + bd)
    W, H = 136, 72
    for it in range(600):
        log2 = int(rng.integers(2, 6)); n = 1 << log2
        c_idx = int(rng.integers(0, 3)); cfi = 1
        sh = 1 if c_idx else 0
        nl = n << sh
        ny = (H - nl) // nl + 1
        x0 = int(rng.integers(0, (W - nl) // nl + 1)) * nl; y0 = int(rng.integers(min(1 if it % 4 else 0, ny - 1), ny)) * nl
        if y0 == 0 and x0 > 0:
            x0 = 0
        mode = int(rng.integers(0, 35))
        cands = [int(rng.random() < 0.8) for _ in range(5)]
        if x0 == 0: cands[0] = cands[1] = cands[2] = 0
        if y0 == 0: cands[2] = cands[3] = cands[4] = 0
        if x0 + nl >= W: cands[4] = 0
        if y0 + nl >= H: cands[0] = 0
        lpu = int(rng.choice([2, 3]))
        pw, ph = (W + (1 << lpu) - 1) >> lpu, (H + (1 << lpu) - 1) >> lpu
        p_intra = float(rng.choice([0.0, 0.2, 0.5, 0.8, 1.0]))
        is_intra = (rng.random((ph, pw)) < p_intra).astype(np.uint8)
        # the block being predicted is intra itself
        is_intra[y0 >> lpu:((y0 + nl - 1) >> lpu) + 1, x0 >> lpu:((x0 + nl - 1) >> lpu) + 1] = 1
        planes = [rand_plane(rng, bd, H + 8, W + 8) for _ in range(3)]
        pa = [p.copy() for p in planes]; pb = [p.copy() for p in planes]
        kw = dict(chroma_format_idc=cfi, strong=int(rng.random() < 0.7), smoothing_disabled=0, log2_ctb_size=6, log2_min_tb_size=2,
                  log2_min_pu_size=lpu, constrained=1, is_intra=is_intra)
        oracle.intra_pred(bd, pa, W, H, x0, y0, log2, c_idx, mode, cands, **kw)
        ref.intra_pred(bd, pb, W, H, x0, y0, log2, c_idx, mode, cands, **kw)
        for i in range(3):
            assert np.array_equal(pa[i], pb[i]), (it, log2, c_idx, mode, cands, x0, y0, lpu, p_intra)


# ------------------------------------------------------------------ SHVC inter-layer up-sampling
def _shvc_case(rng, ratios, use_window):
    bw, bh = int(rng.integers(10, 40)) * 8, int(rng.integers(10, 30)) * 8
    ratio = float(rng.choice(ratios))
    ew, eh = int(round(bw * ratio / 8)) * 8, int(round(bh * ratio / 8)) * 8
    win = (0, 0, 0, 0)
    if use_window:
        win = tuple(int(v) * 2 for v in rng.integers(0, 9, size=4))
    bl = [rand_plane(rng, 8, bh, bw), rand_plane(rng, 8, bh // 2, bw // 2), rand_plane(rng, 8, bh // 2, bw // 2)]
    return bw, bh, ew, eh, win, bl


def _shvc_run(fn, ew, eh):
    el = [np.zeros((eh, ew), np.uint8), np.zeros((eh // 2, ew // 2), np.uint8), np.zeros((eh // 2, ew // 2), np.uint8)]
    _, view = po.padded_planes(el)
    fn(view)
    return [v.copy() for v in view]


def test_shvc_upsample_frame_slot(oracle, ref):
    """upsample_base_layer_frame (hevcdsp_template.c:2165-2438) driven as at hevc.c:3240-3242 (oracle/shvc_driver.c), any ratio,
    with and without a scaled reference layer window.  8 bit: above that the reference's function memsets 16-bit samples
    bytewise at the picture edges (:2233,2243) and corrupts its heap here; SHVC content is 8 bit."""
    rng = np.random.default_rng(310)
    for it in range(40):
        bw, bh, ew, eh, win, bl = _shvc_case(rng, [2.0, 1.5, 1.25, 1.8, 1.0, 3.0], it % 3 == 0)
        up = po.shvc_params(bw, bh, ew, eh, win, phase_align=int(rng.integers(0, 2)))
        _, blv = po.padded_planes(bl)
        a = _shvc_run(lambda v: po.shvc_reference(ref.path, "frame", 8, v, ew, eh, blv, bw, bh, win, up), ew, eh)
        b = _shvc_run(lambda v: po.shvc_upsample_frame(oracle.path, 8, v, ew, eh, blv, bw, bh, win, up, block_slots=0), ew, eh)
        for pl in range(3):
            assert np.array_equal(a[pl], b[pl]), (it, (bw, bh, ew, eh), win, list(up), pl)


def test_shvc_upsample_block_slots(oracle, ref):
    """The shipped path (ACTIVE_PU_UPSAMPLING, hevc.h:117): upsample_filter_block_{luma,cr}_{h,v}[idx] + emulated_edge_up_{h,v}
    called CTB by CTB as upsample_block_luma / upsample_block_mc do (hevc_filter.c:1175-1310, restated in shvc_driver.c),
    idx 0 (general), 1 (x2), 2 (x1.5), with and without phase alignment -- except x1.5 with phase alignment, where the
    reference sizes its source window with the general formula (:1194,1197) but the x1.5 slots read other rows: the first
    row of some CTBs then comes from whatever the previous CTB left in the buffer.  Pictures larger than one CTB (a CTB
    spanning both picture edges skips the bottom emulation, videodsp_template.c:141-151)."""
    rng = np.random.default_rng(311)
    seen = set()
    for it in range(60):
        bw, bh, ew, eh, win, bl = _shvc_case(rng, [2.0, 1.5, 1.25, 1.75], False)
        pa = int(rng.integers(0, 2))
        up = po.shvc_params(bw, bh, ew, eh, win, phase_align=pa)
        if up[8] == po.SHVC_X1_5 and pa:
            continue
        ctb = int(rng.choice([4, 5, 6]))
        _, blv = po.padded_planes(bl)
        a = _shvc_run(lambda v: po.shvc_reference(ref.path, "blocks", 8, v, ew, eh, blv, bw, bh, win, up, log2_ctb=ctb), ew, eh)
        b = _shvc_run(lambda v: po.shvc_upsample_frame(oracle.path, 8, v, ew, eh, blv, bw, bh, win, up, block_slots=1), ew, eh)
        for pl in range(3):
            assert np.array_equal(a[pl], b[pl]), (it, (bw, bh, ew, eh), ctb, list(up), pl)
        seen.add((int(up[8]), pa))
    assert {(0, 0), (0, 1), (1, 0), (1, 1), (2, 0)} <= seen


def test_shvc_x1_5_with_phase_alignment_where_the_reference_is_a_function_of_its_inputs(oracle, ref, monkeypatch):
    """x1.5 with phase alignment through the block slots (the case the test above leaves out).  upsample_block_luma / _mc position and size
    the source window with the general formula, add{X,Y}{Lum,Cr} included (hevc_filter.c:1193-1197, 1259-1263), but the x1.5 slots index
    their rows as 2 (y - top) / 3 (hevcdsp_template.c:2063-2160): where the two disagree the vertical slot reads a row of the thread's
    scratch buffer that no slot call of THIS CTB wrote - whatever an earlier CTB (or nothing) left there, and the decoder up-samples
    CTBs in the order motion compensation happens to reference them (ff_upsample_block, hevc_filter.c:1370-1395).  Shown here by running
    the reference three times with the scratch buffer refilled with a different value in front of every CTB (OHREF_SHVC_POISON[_EACH]):
    the samples that differ between the runs are not a function of the stream.  Every other sample is, and there the restatement must
    equal the reference."""
    rng = np.random.default_rng(313)
    undetermined = determined = 0
    for it in range(24):
        bw, bh = int(rng.integers(5, 20)) * 16, int(rng.integers(5, 15)) * 16          # exactly x1.5
        ew, eh, win = bw * 3 // 2, bh * 3 // 2, (0, 0, 0, 0)
        bl = [rand_plane(rng, 8, bh, bw), rand_plane(rng, 8, bh // 2, bw // 2), rand_plane(rng, 8, bh // 2, bw // 2)]
        up = po.shvc_params(bw, bh, ew, eh, win, phase_align=1)
        assert up[8] == po.SHVC_X1_5
        ctb = int(rng.choice([4, 5, 6]))
        _, blv = po.padded_planes(bl)
        runs = []
        for poison in ("0", "1357", "-2468"):
            monkeypatch.setenv("OHREF_SHVC_POISON", poison)
            monkeypatch.setenv("OHREF_SHVC_POISON_EACH", "1")
            runs.append(_shvc_run(lambda v: po.shvc_reference(ref.path, "blocks", 8, v, ew, eh, blv, bw, bh, win, up, log2_ctb=ctb), ew, eh))
        monkeypatch.delenv("OHREF_SHVC_POISON")
        monkeypatch.delenv("OHREF_SHVC_POISON_EACH")
        ours = _shvc_run(lambda v: po.shvc_upsample_frame(oracle.path, 8, v, ew, eh, blv, bw, bh, win, up, block_slots=1), ew, eh)
        for pl in range(3):
            stable = (runs[0][pl] == runs[1][pl]) & (runs[0][pl] == runs[2][pl])
            # an unprepared scratch row feeds one output row of one CTB: such a row segment is undetermined as a whole (a few of its samples
            # come out equal all the same - clipped to 0 / 255 whatever the scratch held)
            seg = (1 << ctb) >> (1 if pl else 0)
            for x in range(0, stable.shape[1], seg):
                stable[:, x:x + seg] &= stable[:, x:x + seg].all(axis=1, keepdims=True)
            undetermined += int((~stable).sum()); determined += int(stable.sum())
            bad = np.argwhere(stable & (ours[pl] != runs[0][pl]))
            assert bad.size == 0, (it, (bw, bh, ew, eh), ctb, list(up), pl, len(bad), bad[:4].tolist())
    assert undetermined > 0, "the reference never read an unprepared row: the premise of this test is gone"
    assert determined > 20 * undetermined


def test_shvc_reference_above_8_bit_writes_outside_the_picture(ref):
    """Why SHVC above 8 bit is pinned against the restatement only: the reference's vertical slots add the destination stride - a BYTE
    count, frame->linesize[] (hevc_filter.c:1232, 1303) - to a `pixel *` (hevcdsp_template.c:1914, 2017, 2116, 2150): with 16-bit samples
    picture row y lands at row 2y, every other row of the upper half stays unwritten and the lower half goes beyond the plane (in the
    decoder: into the next allocation).  No valid reference output exists there; the product and the oracle keep the 8-bit semantics
    in sample units (tests/test_shvc_gpu.py, bit depths 10 and 14)."""
    rng = np.random.default_rng(314)
    bd, bw, bh = 10, 64, 48
    ew, eh = 128, 96
    up = po.shvc_params(bw, bh, ew, eh, (0, 0, 0, 0), phase_align=0)
    bl = [rng.integers(1, 1 << bd, size=(bh, bw)).astype(np.uint16), rng.integers(1, 1 << bd, size=(bh // 2, bw // 2)).astype(np.uint16),
          rng.integers(1, 1 << bd, size=(bh // 2, bw // 2)).astype(np.uint16)]
    _, blv = po.padded_planes(bl, 256)
    # the enhancement-layer planes sit at the top of buffers three times their height: room for the stray rows
    big = [np.full((3 * eh + 512, ew + 512), 0xFFFF, np.uint16), np.full((3 * eh // 2 + 512, ew // 2 + 512), 0xFFFF, np.uint16),
           np.full((3 * eh // 2 + 512, ew // 2 + 512), 0xFFFF, np.uint16)]
    view = [b[256:, 256:256 + (ew >> (1 if i else 0))] for i, b in enumerate(big)]
    assert po.shvc_reference(ref.path, "blocks", bd, view, ew, eh, blv, bw, bh, (0, 0, 0, 0), up, log2_ctb=5) == 0
    luma = view[0]
    written = (luma != 0xFFFF).any(axis=1)
    assert not written[1:eh:2].any(), "odd rows of the picture were written: the stride is no longer applied twice"
    assert written[0:2 * eh:2].all() and written[eh:2 * eh].any(), "rows beyond the picture's height stayed untouched"


# ------------------------------------------------------------------ boundary strengths
BS_STREAMS = ["ldp_8b", "ldb_8b", "ra_8b_ctb64", "ra_10b_odd", "weighted", "pcm", "cip", "tiles", "tiles_nolf", "slices", "slices_nolf",
              "slices_dep_wpp", "no_tools", "fmt444_8b", "tqb", "ra_8b_foll_leaf", "bqmall_geometry_dense_qp22"]


@pytest.mark.parametrize("name", BS_STREAMS)
def test_boundary_strengths_against_the_reference_front_end(name):
    """ohor_boundary_strengths (hevc_filter.c:584-700, 805-941 restated) against the arrays the reference's own function leaves in
    s->horizontal_bs / vertical_bs, picture by picture, on the committed streams: oracle/null_hooks.c logs every call the reference's front
    end makes (position, size, the slice / tile flags of its CTB) and shadows the motion-field and cbf_luma entries it reads."""
    import os
    from oracle import pystream as ps
    from test_stream_cpu import load_golden
    if not ps.have("null"):
        pytest.skip("oracle/_ref/libopenhevc_null.so not built (needs /root/reference)")
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so")
    aus, _ = load_golden(name)
    seen = dict(pictures=0, calls=0, nonzero=0, strength1=0, motion_only=0)
    with ps.Decoder("null", 1, 1) as d:
        for i, au in enumerate(aus):
            po.bs_tap(d.L, True)
            assert d.L.ohdec_decode(d.h, au, len(au), i + 1) >= 0
            t = po.bs_tap_fetch(d.L)
            if t is None:
                continue                      # parameter sets only, or a picture with deblocking switched off
            v, h = po.boundary_strengths(lib, t["geom"], t["mvf"], t["cbf_luma"], t["calls"], t["n_bs"])
            for mine, theirs, what in ((v, t["vertical_bs"], "vertical"), (h, t["horizontal_bs"], "horizontal")):
                bad = np.flatnonzero(mine != theirs)
                assert bad.size == 0, f"{name} access unit {i} {what}_bs: {bad.size} entries differ, first {bad[:4].tolist()} " \
                                      f"mine {mine[bad[:4]].tolist()} reference {theirs[bad[:4]].tolist()}"
            seen["pictures"] += 1; seen["calls"] += len(t["calls"])
            seen["nonzero"] += int(np.count_nonzero(v) + np.count_nonzero(h)); seen["strength1"] += int(np.count_nonzero(v == 1) + np.count_nonzero(h == 1))
        po.bs_tap(d.L, False)
    assert seen["pictures"] > 0 and seen["calls"] > 0 and seen["nonzero"] > 0, seen
