"""The rest of the per-kernel table (missing 4): the kernel families tools/kernel_rows.py did not time - chroma (epel) motion
compensation, the weighted variants, SAO band offset, chroma deblocking, the small inverse transforms and the DST, SHVC up-sampling, the
coefficient expansion, the motion grid and the boundary strengths.  Same rules as kernel_rows.py: HBM-resident rings of at least 1 GiB, time =
median of HIP-event bursts on the launch stream, `achieved` = algorithmic bytes (SURVEY 8d per-unit figures) / time, `frac` = achieved / 8 TB/s,
and every row with a sampled bit-exact check against the CPU oracle (oracle/liboracle.so: test infrastructure, the checker only)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openhevc_amd import lib as L  # noqa: E402
import kernel_rows as K  # noqa: E402

W, H = K.W, K.H
N_CHECK = K.N_CHECK
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")


def _window(ref, sx, sy, w, h):
    return K._window(ref, sx, sy, w, h)


def mc_more_rows(bd, orc, po, g, rng, st, out):
    """epel (chroma) uni / bi and the weighted luma variants (hevcdsp_template.c:985-1174 weighted, 1179-1609 epel)"""
    P = 2 if bd > 8 else 1
    #      name                      plane bw  bh  bi  weighted small
    for name, plane, bw, bh, bi, wt, small in (("mc_chroma_4x4_uni", 1, 4, 4, 0, 0, True), ("mc_chroma_8x8_bi", 1, 8, 8, 1, 0, True),
                                               ("mc_luma_8x8_uni_w", 0, 8, 8, 0, 1, True), ("mc_luma_16x16_bi_w", 0, 16, 16, 1, 1, False)):
        pw, ph = (W, H) if plane == 0 else (W // 2, H // 2)
        xs, ys = np.meshgrid(np.arange(0, pw - bw + 1, bw), np.arange(0, ph - bh + 1, bh))
        n = xs.size
        j = np.zeros(n, L.MC_JOB)
        j["x"], j["y"], j["w"], j["h"], j["plane"] = xs.ravel(), ys.ravel(), bw, bh, plane
        j["flags"] = (L.MC_BI if bi else 0) | (L.MC_WEIGHTED if wt else 0)
        frac = 4 if plane == 0 else 8
        for s in ("0", "1"):
            j["sx" + s] = j["x"].astype(np.int32) + rng.integers(-16, 17, n)
            j["sy" + s] = j["y"].astype(np.int32) + rng.integers(-16, 17, n)
            j["mx" + s], j["my" + s] = rng.integers(0, frac, n), rng.integers(0, frac, n)
            j["wx" + s], j["ox" + s] = rng.integers(-64, 128, n), rng.integers(-32, 32, n)
        j["denom"] = rng.integers(0, 8, n)
        j["ref1"] = 1
        d_jobs = K._dev(j)

        def sources(k):
            rr = [K._smooth_pic(bd, g) for _ in range(2)]
            return rr, K._dev(L.planes_table(rr))

        def launch(pic, ex):
            if small:
                L.dev_mc_batch_small(L.planes_of(pic), ex[1].data_ptr(), 2, bd, d_jobs.data_ptr(), n, st())
            else:
                L.dev_mc_batch_bounded(L.planes_of(pic), ex[1].data_ptr(), 2, bd, d_jobs.data_ptr(), n, bw, bh, st())
        src_bytes = 2 * K._bytes(K._smooth_pic(bd, g))
        ms, ring = K._time(launch, lambda: K._smooth_pic(bd, g), sources, src_bytes)
        pic, ex = K._smooth_pic(bd, g), sources(0)
        launch(pic, ex)
        torch.cuda.synchronize()
        got, r0, r1 = K._np(pic[plane], bd), K._np(ex[0][0][plane], bd), K._np(ex[0][1][plane], bd)
        bad = 0
        luma = plane == 0
        for k in rng.integers(0, n, N_CHECK):
            q = j[k]
            x, y = int(q["x"]), int(q["y"])
            kw = dict(denom=int(q["denom"]), wx0=int(q["wx0"]), wx1=int(q["wx1"]), ox0=int(q["ox0"]), ox1=int(q["ox1"]))
            if not bi:
                want = orc.mc(bd, luma, po.MC_UNI_W if wt else po.MC_UNI, _window(r0, int(q["sx0"]), int(q["sy0"]), bw, bh), 3, 3, bw, bh, int(q["mx0"]), int(q["my0"]), **kw)
            else:
                tmp = orc.mc(bd, luma, po.MC_PUT, _window(r0, int(q["sx0"]), int(q["sy0"]), bw, bh), 3, 3, bw, bh, int(q["mx0"]), int(q["my0"]))
                src2 = np.zeros((bh, 64), np.int16)
                src2[:, :bw] = tmp
                want = orc.mc(bd, luma, po.MC_BI_W if wt else po.MC_BI, _window(r1, int(q["sx1"]), int(q["sy1"]), bw, bh), 3, 3, bw, bh, int(q["mx1"]), int(q["my1"]), src2=src2, **kw)
            bad += not np.array_equal(got[y:y + bh, x:x + bw], want)
        T = 8 if luma else 4
        alg = n * ((1 + bi) * P * (bw + T - 1) * (bh + T - 1) + P * bw * bh)
        out[f"{name}_{bd}bit"] = K._row(
            ms, ring, alg, n * bw * bh, bad, N_CHECK,
            f"{n} {'luma' if luma else 'chroma (4:2:0 Cb plane)'} blocks {bw}x{bh}, {'bi' if bi else 'uni'}-prediction{', explicit weights (random denom / weights / offsets)' if wt else ''}, "
            f"random fractional phases and vectors within +-16 samples, {'small-block' if small else 'tile'} entry point; put_hevc_{'qpel' if luma else 'epel'}_"
            f"{'bi' if bi else 'uni'}{'_w' if wt else ''}_* (hevcdsp_template.c:{'985-1174' if wt else '1179-1609'})")


def sao_band_rows(bd, orc, po, g, rng, st, out):
    P = 2 if bd > 8 else 1
    xs, ys = np.meshgrid(np.arange(0, W, 64), np.arange(0, H, 64))
    n = xs.size
    j = np.zeros(n, L.SAO_JOB)
    j["x"], j["y"] = xs.ravel(), ys.ravel()
    j["w"], j["h"] = np.minimum(64, W - j["x"]), np.minimum(64, H - j["y"])
    j["type"] = L.SAO_BAND
    j["klass"] = rng.integers(0, 32, n)                                   # band_position (sao_band_filter's left class)
    j["borders"] = (j["x"] == 0) * 1 + (j["y"] == 0) * 2 + (j["x"] + j["w"] == W) * 4 + (j["y"] + j["h"] == H) * 8
    j["offset_val"] = [0, 3, 1, -1, -3]
    d_jobs = K._dev(j)

    def launch(pic, ex):
        L.dev_sao_batch_sorted(L.planes_of(pic), L.planes_of(ex), bd, d_jobs.data_ptr(), n, 0, st())
    ms, ring = K._time(launch, lambda: K._smooth_pic(bd, g), lambda k: K._smooth_pic(bd, g), K._bytes(K._smooth_pic(bd, g)))
    pic, src = K._smooth_pic(bd, g), K._smooth_pic(bd, g)
    launch(pic, src)
    torch.cuda.synchronize()
    got, s0 = K._np(pic[0], bd), K._np(src[0], bd)
    bad = 0
    want = np.zeros_like(s0)
    for k in rng.integers(0, n, N_CHECK):
        x, y, w, h = int(j["x"][k]), int(j["y"][k]), int(j["w"][k]), int(j["h"][k])
        orc.sao_band(bd, want, s0, x, y, w, h, [0, 3, 1, -1, -3], int(j["klass"][k]))
        bad += not np.array_equal(got[y:y + h, x:x + w], want[y:y + h, x:x + w])
    out[f"sao_band_luma_{bd}bit"] = K._row(
        ms, ring, 2 * P * W * H, W * H, bad, N_CHECK,
        f"{n} luma CTBs 64x64, band offset with a random band position per CTB, reading a deblocked copy and writing the picture; sao_band_filter (hevcdsp_template.c:340-365)")


def deblock_chroma_rows(bd, orc, po, g, rng, st, out):
    """chroma edges are filtered where bS = 2 on the 16-sample luma grid = the 8-sample chroma grid of 4:2:0 (hevc_filter.c:470-580,
    hevc_{h,v}_loop_filter_chroma: hevcdsp_template.c:1725-1771): bS 2 on every 8x8-grid edge, QP 38, the luma planes of the ring are filtered too
    (one launch does all three planes); the row's bytes count all three planes"""
    P = 2 if bd > 8 else 1
    bw, bh = W >> 2, H >> 2
    vb = np.zeros(bw * (bh + 8), np.uint8)
    hb = np.zeros((bw + 8) * bh, np.uint8)
    grid = np.zeros((bh, bw), np.uint8)
    grid[:, ::2] = 2
    vb[:bw * bh] = grid.ravel()
    hgrid = np.zeros((bh, bw), np.uint8)
    hgrid[::2, :] = 2
    hb[:bw * bh] = hgrid.ravel()
    qp_y = 38
    qp = np.full((W >> 3) * (H >> 3), qp_y, np.int8)
    dbp = np.zeros((((W + 63) // 64) * ((H + 63) // 64), 2), np.int8)
    keep = [K._dev(a) for a in (vb, hb, qp, dbp)]
    dm = L.DbkMaps(vertical_bs=keep[0].data_ptr(), horizontal_bs=keep[1].data_ptr(), qp_y_tab=keep[2].data_ptr(), deblock=keep[3].data_ptr(), is_pcm=None,
                   bs_width=bw, min_cb_width=W >> 3, deblock_stride=2, min_pu_width=W >> 2, min_pu_height=H >> 2, width=W, height=H, log2_ctb_size=6,
                   log2_min_cb_size=3, log2_min_pu_size=2, chroma_format_idc=1, cb_qp_offset=0, cr_qp_offset=0)
    # chroma QP of luma QP 38 at 4:2:0: table 8-10 (hevc_filter.c:118-142: qp_c[] for qPi 30..43) -> 35; tc = tctable[QPc + 2 (bS 2)] (hevc_filter.c:62-89, 520-527)
    qpc = [29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37][qp_y - 30]
    tc = K.TC_TABLE[qpc + 2]
    for vertical, word in ((1, "vertical"), (0, "horizontal")):
        def launch(pic, ex):
            L.dev_deblock_maps(L.planes_of(pic), bd, dm, vertical, st())
        ms, ring = K._time(launch, lambda: K._smooth_pic(bd, g))
        pic = K._smooth_pic(bd, g)
        before = K._np(pic[1], bd).copy()
        launch(pic, None)
        torch.cuda.synchronize()
        got = K._np(pic[1], bd)
        bad = 0
        cw, chh = W // 2, H // 2
        for _ in range(N_CHECK):
            if vertical:
                x, y = 8 * int(rng.integers(1, cw // 8)), 8 * int(rng.integers(0, chh // 8))
                want = before[y:y + 8, x - 8:x + 8].copy()
                for half in (0, 4):                            # the reference filters a chroma edge in pieces of 4 rows x 2 (tc per piece): one call = 8 rows
                    pass
                orc.deblock_chroma(bd, 1, want, 8, 0, (tc, tc), (0, 0), (0, 0))
                bad += not np.array_equal(got[y:y + 8, x - 4:x + 4], want[:, 4:12])
            else:
                x, y = 8 * int(rng.integers(0, cw // 8)), 8 * int(rng.integers(1, chh // 8))
                want = before[y - 8:y + 8, x:x + 8].copy()
                orc.deblock_chroma(bd, 0, want, 0, 8, (tc, tc), (0, 0), (0, 0))
                bad += not np.array_equal(got[y - 4:y + 4, x:x + 8], want[4:12, :])
        # touched: every luma sample (8x8 grid, 4 samples either side ... the luma row's figure) + every chroma sample within 1 of an 8-grid edge is
        # read+written: SURVEY 8d "2P bytes per touched pixel", frame bound 2P x 1.5 W H
        out[f"deblock_all_planes_{word}_bs2_from_maps_{bd}bit"] = K._row(
            ms, ring, 2 * P * W * H * 3 // 2, W * H * 3 // 2, bad, N_CHECK,
            f"every {word} 8x8-grid edge of eight stacked 4K 4:2:0 pictures with bS 2: luma AND both chroma planes in one launch (chroma edges on their 8-sample "
            f"grid), parameters derived on the device from the decoder's maps (QP 38 -> QPc 35); checked on the Cb plane; hevc_{'v' if vertical else 'h'}_loop_filter_chroma "
            f"(hevcdsp_template.c:1725-1771), chroma QP mapping hevc_filter.c:118-142")


def tu_small_rows(bd, orc, po, g, rng, st, out):
    P = 2 if bd > 8 else 1
    for log2, kind, okind, name, cite in ((2, L.TU_IDCT, po.TU_IDCT, "idct_add_4x4", "idct_4x4 (hevcdsp_template.c:210-262)"),
                                          (2, L.TU_DST4, po.TU_DST4, "dst_add_4x4", "transform_4x4_luma (hevcdsp_template.c:170-203)"),
                                          (3, L.TU_IDCT, po.TU_IDCT, "idct_add_8x8", "idct_8x8 (hevcdsp_template.c:210-262)")):
        nn = 1 << log2
        xs, ys = np.meshgrid(np.arange(0, W, nn), np.arange(0, H, nn))
        n = xs.size
        j = np.zeros(n, L.TU_JOB)
        j["x"], j["y"], j["coeff_off"] = xs.ravel(), ys.ravel(), np.arange(n, dtype=np.uint32) * nn * nn
        d_jobs = K._dev(j)
        coeffs = torch.randint(-1024, 1024, (n, nn, nn), dtype=torch.int16, device="cuda", generator=g)
        dt = torch.uint8 if bd == 8 else torch.int16

        def fresh():
            return [torch.randint(0, 1 << bd, (H, W), dtype=torch.int32, device="cuda", generator=g).to(dt), None, None]

        def launch(pic, ex):
            L.dev_tu_batch(L.planes_of(pic), bd, log2, kind, d_jobs.data_ptr(), n, coeffs.data_ptr(), st())
        ms, ring = K._time(launch, fresh)
        pic = fresh()
        picks = [(int(k), K._np(pic[0][int(j["y"][k]):int(j["y"][k]) + nn, int(j["x"][k]):int(j["x"][k]) + nn], bd).copy()) for k in rng.integers(0, n, N_CHECK)]
        launch(pic, None)
        torch.cuda.synchronize()
        bad = 0
        for k, before in picks:
            x, y = int(j["x"][k]), int(j["y"][k])
            want = orc.tu_batch(bd, okind, log2, coeffs[k:k + 1].cpu().numpy(), before, np.zeros((1, 2), np.int32))
            bad += not np.array_equal(K._np(pic[0][y:y + nn, x:x + nn], bd), want)
        out[f"{name}_{bd}bit"] = K._row(
            ms, ring + coeffs.numel() * 2, n * nn * nn * (2 + 2 * P), n * nn * nn, bad, N_CHECK,
            f"{n} blocks {nn}x{nn} tiling eight stacked 4K luma planes, {bd}-bit, coefficients U[-1024,1023]; {cite} + transform_add")
        del coeffs, d_jobs, pic


def upsample_rows(bd, orc, po, g, rng, st, out):
    """SHVC inter-layer up-sampling, luma plane, x2 and x1.5 (hevcdsp_template.c:1807-2444: upsample_filter_block_luma_h / _v)"""
    P = 2 if bd > 8 else 1
    dt = torch.uint8 if bd == 8 else torch.int16
    ndt = np.uint16 if bd > 8 else np.uint8
    for label, num, den in (("x2", 2, 1), ("x1_5", 3, 2)):
        bw, bh = W * den // num // 16 * 16, H * den // num // 16 * 16
        ew, eh = bw * num // den, bh * num // den
        win = (0, 0, 0, 0)
        up = po.shvc_params(bw, bh, ew, eh, win, phase_align=0)
        prm = L.upsample_params(ew, eh, bw, bh, win, up, 0)
        cols, col_of, rows, sc, sr = L.upsample_maps(prm, 0)
        keep = [K._dev(a) for a in (cols, col_of, rows)]

        def fresh():
            return [torch.zeros((eh, ew), dtype=dt, device="cuda"), None, None]

        def base(k):
            return torch.randint(0, 1 << bd, (bh, bw), dtype=torch.int32, device="cuda", generator=g).to(dt)

        def launch(pic, ex):
            L.dev_upsample_plane(pic[0], ex, bd, 0, keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), sc, sr, st())
        ms, ring = K._time(launch, fresh, base, bw * bh * P)
        # check: a small picture of the same ratio through the same entry point against the oracle's frame function (the 4K-high plane would take
        # the plain-C restatement minutes)
        sbw, sbh = 16 * 12, 16 * 8
        sew, seh = sbw * num // den, sbh * num // den
        sup = po.shvc_params(sbw, sbh, sew, seh, win, phase_align=0)
        sprm = L.upsample_params(sew, seh, sbw, sbh, win, sup, 0)
        scols, scol_of, srows, ssc, ssr = L.upsample_maps(sprm, 0)
        skeep = [K._dev(a) for a in (scols, scol_of, srows)]
        bl = [rng.integers(0, 1 << bd, size=(sbh, sbw)).astype(ndt), rng.integers(0, 1 << bd, size=(sbh // 2, sbw // 2)).astype(ndt),
              rng.integers(0, 1 << bd, size=(sbh // 2, sbw // 2)).astype(ndt)]
        want = [np.zeros((seh, sew), ndt), np.zeros((seh // 2, sew // 2), ndt), np.zeros((seh // 2, sew // 2), ndt)]
        po.shvc_upsample_frame(ORACLE_SO, bd, want, sew, seh, bl, sbw, sbh, win, sup, block_slots=0)
        d_src = torch.from_numpy(bl[0].view(np.int16) if bd > 8 else bl[0]).cuda()
        d_dst = torch.zeros((seh, sew), dtype=dt, device="cuda")
        L.dev_upsample_plane(d_dst, d_src, bd, 0, skeep[0].data_ptr(), skeep[1].data_ptr(), skeep[2].data_ptr(), ssc, ssr, st())
        torch.cuda.synchronize()
        got = K._np(d_dst, bd)
        units = (seh // 16) * (sew // 16)
        badu = sum(not np.array_equal(got[y:y + 16, x:x + 16], want[0][y:y + 16, x:x + 16]) for y in range(0, seh, 16) for x in range(0, sew, 16))
        out[f"shvc_upsample_luma_{label}_{bd}bit"] = K._row(
            ms, ring, P * ew * eh + P * bw * bh, ew * eh, badu, units,
            f"base-layer luma plane {bw}x{bh} -> {ew}x{eh} ({label.replace('_', '.')}), general 8-tap filter rules, one launch per plane; check: a {sbw}x{sbh} plane of the same "
            f"ratio through the same entry point against the oracle's frame function, all {units} 16x16 units; upsample_filter_block_luma_h / _v (hevcdsp_template.c:1807-2444)")


def expand_rows(orc, po, g, rng, st, out):
    """ohevc_dev_expand_coeffs: the compact coefficient stream (the col_limit rectangle of every inverse-DCT block, hevcdsp.h:53's argument) back into the
    dense arena.  Encoder-like mix of sizes and rectangles; bytes = compact read + records + dense write (zeros included: the arena is written whole)."""
    nblk = 1 << 19
    log2 = rng.choice([2, 3, 4, 5], nblk, p=[0.35, 0.35, 0.2, 0.1]).astype(np.int64)
    n = 1 << log2
    lim = np.minimum(n, 4 * rng.integers(1, 9, nblk))                    # cols: a multiple of 4 up to N
    rows = np.minimum(n, lim + 4 * rng.integers(0, 2, nblk))
    whole = (log2 < 3) | ((lim == n) & (rows == n))
    cols = np.where(whole, n, lim)
    rows = np.where(whole, n, rows)
    src_len = np.where(whole, n * n, cols * rows)
    src = np.concatenate([[0], np.cumsum(src_len)[:-1]])
    dst = np.concatenate([[0], np.cumsum(n * n)[:-1]])
    rec = np.zeros(nblk, np.dtype([("src", "<u4"), ("dst", "<u4"), ("dims", "<u4"), ("kind", "<u4")]))
    rec["src"], rec["dst"] = src, dst
    rec["dims"] = np.where(whole, n * n, cols | (rows << 8))
    rec["kind"] = np.where(whole, 0, log2)
    total_src, total_dst = int(src_len.sum()), int((n * n).sum())
    compact = torch.randint(-1024, 1024, (total_src,), dtype=torch.int16, device="cuda", generator=g)
    d_rec = K._dev(rec)
    lib = L.load_library()
    import ctypes as C

    def fresh():
        return [torch.full((total_dst,), 0x5a5a, dtype=torch.int16, device="cuda"), None, None]

    def launch(pic, ex):
        L.check(lib.ohevc_dev_expand_coeffs(C.c_void_p(compact.data_ptr()), C.c_void_p(d_rec.data_ptr()), C.c_int(nblk), C.c_void_p(pic[0].data_ptr()), C.c_void_p(st())))
    ms, ring = K._time(launch, fresh)
    dense = fresh()
    launch(dense, None)
    torch.cuda.synchronize()
    got, cpt = dense[0].cpu().numpy(), compact.cpu().numpy()
    bad = 0
    for k in rng.integers(0, nblk, N_CHECK * 4):
        nn = int(n[k])
        want = np.zeros((nn, nn), np.int16)
        if whole[k]:
            want[...] = cpt[src[k]:src[k] + nn * nn].reshape(nn, nn)
        else:
            want[:rows[k], :cols[k]] = cpt[src[k]:src[k] + cols[k] * rows[k]].reshape(rows[k], cols[k])
        bad += not np.array_equal(got[dst[k]:dst[k] + nn * nn].reshape(nn, nn), want)
    alg = 2 * total_src + 16 * nblk + 2 * total_dst
    out["expand_coeffs"] = K._row(
        ms, ring + 2 * total_src, alg, total_dst, bad, N_CHECK * 4,
        f"{nblk} transform blocks (4x4 35 %, 8x8 35 %, 16x16 20 %, 32x32 10 %), random col_limit rectangles, {total_src * 2 >> 20} MiB compact -> {total_dst * 2 >> 20} MiB dense arena; "
        f"the device half of ohevc_rec_tu_limited (the col_limit argument of hevcdsp.h:53; read pattern of idct_full, hevcdsp_template.c:271-291); `mpixel_per_s` = coefficients/s")


def bs_rows(orc, po, g, rng, st, out):
    """ohevc_dev_motion_grid2 + ohevc_dev_boundary_strengths on one synthetic 1080p picture (random coding / prediction / transform trees: the generator of
    tests/test_boundary_strength_gpu.py), all entries compared with the oracle's restatement of hevc_filter.c:584-700,805-941"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_boundary_strength_gpu as T
    Wp, Hp, log2_ctb, l2pu = 1920, 1080 // 8 * 8, 6, 2
    field, cbf, calls, pus = T.synth_picture(rng, Wp, Hp, log2_ctb, l2pu)
    pu_w, pu_h, tb_w, tb_h = Wp >> l2pu, Hp >> l2pu, Wp >> 2, Hp >> 2
    geom = dict(min_pu_width=pu_w, log2_min_pu_size=l2pu, min_tb_width=tb_w, log2_min_tb_size=2, log2_ctb_size=log2_ctb, bs_width=Wp >> 2, loop_filter_across_tiles=1)
    n_bs = (Wp >> 2) * (Hp >> 2)
    want = po.boundary_strengths(ORACLE_SO, geom, field, cbf, calls.view(po.BS_CALL), n_bs)
    jobs = T.jobs_of_pus(rng, pus)
    jobs = jobs[jobs["plane"] == 0]
    d_jobs, d_cbf, d_calls = K._dev(jobs), K._dev(cbf), K._dev(calls)
    # several pictures' worth of buffers so that a launch does not find its inputs in the cache of the launch before (a picture is small: 2.6 MB of grid)
    NB = 64
    grids = [torch.zeros(pu_w * pu_h * L.MOTION_GRID_ENTRY, dtype=torch.uint8, device="cuda") for _ in range(NB)]
    bsv = [torch.zeros(n_bs, dtype=torch.uint8, device="cuda") for _ in range(NB)]
    bsh = [torch.zeros(n_bs, dtype=torch.uint8, device="cuda") for _ in range(NB)]
    stream = torch.cuda.current_stream()

    def burst(fn, reps=6):
        ts = []
        for r in range(reps + 1):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for k in range(NB):
                fn(k)
            b.record(stream)
            torch.cuda.synchronize()
            if r:
                ts.append(a.elapsed_time(b) / NB)
        return float(np.median(ts))
    ms_grid = burst(lambda k: L.dev_motion_grid2(d_jobs.data_ptr(), len(jobs), 0, 0, grids[k].data_ptr(), pu_w, pu_h, l2pu, st()))
    got_grid = grids[0].cpu().numpy().view(po.BS_FIELD)
    bad_grid = int(np.count_nonzero((got_grid["pred_flag"] != 0) != (field["pred_flag"] != 0)))

    def run_bs(k):
        maps = L.BsMaps(mvf=grids[k].data_ptr(), mvf_stride=20, off_mv=0, off_poc=8, off_pred_flag=16, pred_flag_bytes=4, cbf_luma=d_cbf.data_ptr(),
                        min_pu_width=pu_w, min_pu_height=pu_h, log2_min_pu_size=l2pu, min_tb_width=tb_w, min_tb_height=tb_h, log2_min_tb_size=2,
                        log2_ctb_size=log2_ctb, bs_width=Wp >> 2, width=Wp, height=Hp, loop_filter_across_tiles=1)
        L.dev_boundary_strengths(maps, d_calls.data_ptr(), len(calls), bsv[k].data_ptr(), bsh[k].data_ptr(), st())
    ms_bs = burst(run_bs)
    gv, gh = bsv[1].cpu().numpy(), bsh[1].cpu().numpy()
    bad_bs = int(np.count_nonzero(gv != want[0]) + np.count_nonzero(gh != want[1]))
    ring = NB * (grids[0].numel() + 2 * n_bs)
    out["motion_grid_1080p"] = K._row(
        ms_grid, ring, len(jobs) * 32 + pu_w * pu_h * L.MOTION_GRID_ENTRY, Wp * Hp, bad_grid, pu_w * pu_h,
        f"{len(jobs)} luma MC jobs of one synthetic {Wp}x{Hp} picture scattered into the {L.MOTION_GRID_ENTRY}-byte-per-4x4 motion grid (what the boundary strengths read instead of "
        f"an uploaded tab_mvf); bytes = jobs read + grid written; checked: exactly the inter-predicted units are written (all {pu_w * pu_h})")
    out["boundary_strengths_1080p"] = K._row(
        ms_bs, ring, len(calls) * 8 + pu_w * pu_h * L.MOTION_GRID_ENTRY + tb_w * tb_h + 2 * n_bs, Wp * Hp, bad_bs, 2 * n_bs,
        f"{len(calls)} ff_hevc_deblocking_boundary_strengths calls of one synthetic {Wp}x{Hp} picture (random coding / prediction / transform trees) from the motion grid; "
        f"bytes = call records + grid + cbf map read, both bS arrays written; all {2 * n_bs} entries compared with the oracle (hevc_filter.c:584-700,805-941)")


def run(orc, po, out, only=None):
    g = torch.Generator(device="cuda").manual_seed(11)
    rng = np.random.default_rng(11)
    st = lambda: torch.cuda.current_stream().cuda_stream
    fams = []
    for bd in (8, 10):
        fams += [("mc_more", lambda bd=bd: mc_more_rows(bd, orc, po, g, rng, st, out)), ("sao_band", lambda bd=bd: sao_band_rows(bd, orc, po, g, rng, st, out)),
                 ("deblock_chroma", lambda bd=bd: deblock_chroma_rows(bd, orc, po, g, rng, st, out)), ("tu_small", lambda bd=bd: tu_small_rows(bd, orc, po, g, rng, st, out)),
                 ("upsample", lambda bd=bd: upsample_rows(bd, orc, po, g, rng, st, out))]
    fams += [("expand", lambda: expand_rows(orc, po, g, rng, st, out)), ("bs", lambda: bs_rows(orc, po, g, rng, st, out))]
    for name, fn in fams:
        if only is None or only in name:
            try:
                fn()
            except Exception as e:                       # a row that cannot run must not take the bench line with it
                out[f"{name}_error"] = f"{type(e).__name__}: {e}"
            torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    import json
    from oracle import pyoracle as po
    res = run(po.load("oracle"), po, {}, sys.argv[1] if len(sys.argv) > 1 else None)
    for k, v in res.items():
        print(k, json.dumps(v)[:400])
