#!/usr/bin/env python3
"""Seeded synthetic *job streams*: random but LEGAL HEVC reconstruction work tiling a W x H 4:2:0 picture.

No HEVC bitstream, encoder or conformance stream exists in this environment, so configs that call
for real streams are exercised with the call sequence a decoder WOULD make for a picture: CTUs in raster order, CUs in
z-order, per CU either inter prediction (uni/bi, optional weighting, MVs possibly pointing outside the picture) followed
by its residual TUs, or intra prediction TU by TU in z-order with neighbour availability exactly as
ff_hevc_set_neighbour_available computes it (hevc_mvs.c:41-58); then deblocking edges on the 8x8 grid (all vertical,
then all horizontal: hevc_filter.c:385-580) and one SAO job per CTB and plane.

The same op list drives (a) the CPU oracle, applied strictly in this decode order, and (b) the GPU ctx layer, which is
free to reorder it into phases -- comparing the two validates the executor's ordering rules, not just the kernels.
This is test/bench infrastructure: it imports nothing from oracle/ and nothing from the product.
"""
import numpy as np

TU_IDCT, TU_DC, TU_DST4, TU_SKIP, TU_BYPASS = 0, 1, 2, 3, 6


def gen_frame_ops(rng, W, H, bd, n_refs=2, intra_frac=0.15, bi_frac=0.6, coded_frac=0.4, weighted_frac=0.1, log2_ctb=6,
                  pcm_frac=0.01):
    """Returns (ops, filter_ops).  Coordinates of 'tu'/'intra' ops are LUMA positions + c_idx like the reference's calls."""
    ops = []
    ctb = 1 << log2_ctb

    def coeffs_for(n, kind):
        c = np.zeros((n, n), np.int16)
        if kind == TU_DC:
            c[0, 0] = rng.integers(-2000, 2000)
            return c
        k = int(rng.choice([2, 4, 8])) if n > 4 else 4
        k = min(k, n)
        c[:k, :k] = rng.integers(-600, 600, size=(k, k))
        if rng.random() < 0.1:
            c = rng.integers(-1024, 1024, size=(n, n)).astype(np.int16)
        return c

    def residual(x0, y0, log2, intra, c_idx):
        n = 1 << log2
        r = rng.random()
        kind = TU_DC if r < 0.25 else (TU_DST4 if (intra and c_idx == 0 and log2 == 2) else TU_IDCT)
        if r > 0.97:
            kind = TU_SKIP if log2 == 2 else TU_BYPASS
        ops.append(dict(t="tu", c_idx=c_idx, x0=x0, y0=y0, log2=log2, kind=kind, coeffs=coeffs_for(n, kind), intra=intra))

    for cty in range(0, H, ctb):
        for ctx_ in range(0, W, ctb):
            def cu(x0, y0, log2):
                size = 1 << log2
                if x0 >= W or y0 >= H:
                    return
                if log2 > 3 and (x0 + size > W or y0 + size > H or rng.random() < (0.75 if log2 > 4 else 0.45)):
                    h = size >> 1
                    for (dx, dy) in ((0, 0), (h, 0), (0, h), (h, h)):
                        cu(x0 + dx, y0 + dy, log2 - 1)
                    return
                if x0 + size > W or y0 + size > H:
                    return
                if log2 <= 5 and rng.random() < pcm_frac:
                    pcm_cu(x0, y0, log2)
                elif rng.random() < intra_frac:
                    intra_cu(x0, y0, log2)
                else:
                    inter_cu(x0, y0, log2)

            def pcm_cu(x0, y0, log2):
                # hls_pcm_sample (hevc.c:1587-1621): raw samples for the three planes, pcm bit depth <= bit depth
                for c_idx in range(3):
                    l2 = log2 - (1 if c_idx else 0)
                    pcm_bd = int(rng.integers(4, bd + 1))
                    n = 1 << l2
                    ops.append(dict(t="pcm", c_idx=c_idx, x0=x0, y0=y0, log2=l2, pcm_bd=pcm_bd, bd=bd,
                                    samples=rng.integers(0, 1 << pcm_bd, size=(n, n))))

            def inter_cu(x0, y0, log2):
                size = 1 << log2
                parts = [(0, 0, size, size)]
                if size >= 16 and rng.random() < 0.3:
                    parts = [(0, 0, size // 2, size), (size // 2, 0, size // 2, size)] if rng.random() < 0.5 else \
                            [(0, 0, size, size // 4), (0, size // 4, size, 3 * size // 4)]
                for (px, py, pw, ph) in parts:
                    bi = rng.random() < bi_frac and not (pw + ph == 12)
                    wt = rng.random() < weighted_frac
                    mv = [(int(rng.integers(-64, 65)), int(rng.integers(-64, 65))) for _ in range(2)]
                    if rng.random() < 0.05:
                        mv[0] = (int(rng.integers(-4 * W, 4 * W)), int(rng.integers(-4 * H, 4 * H)))
                    ops.append(dict(t="mc", x0=x0 + px, y0=y0 + py, w=pw, h=ph, bi=bi, weighted=wt, mv=mv,
                                    ref=[int(rng.integers(0, n_refs)), int(rng.integers(0, n_refs))],
                                    denom=int(rng.integers(0, 8)), wx=[int(rng.integers(-64, 128)), int(rng.integers(-64, 128))],
                                    ox=[int(rng.integers(-32, 32)), int(rng.integers(-32, 32))]))
                if rng.random() < coded_frac:
                    tlog = min(log2, 5)
                    if log2 > 2 and rng.random() < 0.4:
                        tlog = log2 - 1
                    tn = 1 << tlog
                    for ty in range(y0, y0 + size, tn):
                        for tx in range(x0, x0 + size, tn):
                            if rng.random() < 0.7:
                                residual(tx, ty, tlog, 0, 0)
                            if tlog > 2:
                                for c_idx in (1, 2):
                                    if rng.random() < 0.5:
                                        residual(tx, ty, tlog - 1, 0, c_idx)
                    if tlog == 2:                       # 8x8 CU with four 4x4 luma TUs: chroma once, at the CU origin
                        for c_idx in (1, 2):
                            if rng.random() < 0.5:
                                residual(x0, y0, 2, 0, c_idx)

            def intra_cu(x0, y0, log2):
                size = 1 << log2
                tlog = min(log2, 5)
                if log2 > 2 and rng.random() < 0.5:
                    tlog = log2 - 1
                if log2 == 3 and rng.random() < 0.5:
                    tlog = 2
                tn = 1 << tlog
                mode_c = int(rng.choice([0, 1, 10, 26, int(rng.integers(2, 35))]))
                order = [(tx, ty) for ty in range(0, size, tn) for tx in range(0, size, tn)]
                if len(order) == 4:
                    order = [(0, 0), (tn, 0), (0, tn), (tn, tn)]
                elif len(order) == 16:                  # z-order of a 4x4 arrangement
                    order = [((i & 1) + ((i >> 2) & 1) * 2, ((i >> 1) & 1) + ((i >> 3) & 1) * 2) for i in range(16)]
                    order = [(a * tn, b * tn) for a, b in order]
                for k, (tx, ty) in enumerate(order):
                    mode = int(rng.integers(0, 35))
                    intra_tu(x0 + tx, y0 + ty, tlog, 0, mode)
                    if rng.random() < 0.6:
                        residual(x0 + tx, y0 + ty, tlog, 1, 0)
                    if tlog > 2:
                        for c_idx in (1, 2):
                            intra_tu(x0 + tx, y0 + ty, tlog - 1, c_idx, mode_c)
                            if rng.random() < 0.5:
                                residual(x0 + tx, y0 + ty, tlog - 1, 1, c_idx)
                    elif k == len(order) - 1 or (len(order) == 16 and k % 4 == 3):
                        # 4x4 luma TUs: chroma 4x4 after the 4th luma block of each 8x8 (blk_idx == 3, hevc.c:1364-1394)
                        bx, by = (x0 + tx) & ~7, (y0 + ty) & ~7
                        for c_idx in (1, 2):
                            intra_tu(bx, by, 2, c_idx, mode_c, nb_size=8)
                            if rng.random() < 0.5:
                                residual(bx, by, 2, 1, c_idx)

            def intra_tu(x0, y0, log2, c_idx, mode, nb_size=None):
                # ff_hevc_set_neighbour_available(s, x0, y0, nPbW, nPbH) with the TU size in luma samples (hevc_mvs.c:41-58)
                n_l = nb_size if nb_size is not None else ((1 << log2) << (1 if c_idx else 0))
                x0b, y0b = x0 & (ctb - 1), y0 & (ctb - 1)
                ctb_left, ctb_up = ctx_ > 0, cty > 0
                ctb_up_left = ctb_left and ctb_up
                ctb_up_right = ctb_up and (ctx_ + ctb) < W
                cand_up = bool(ctb_up or y0b)
                cand_left = bool(ctb_left or x0b)
                cand_up_left = ctb_up_left if (not x0b and not y0b) else (cand_left and cand_up)
                sap = (ctb_up_right and not y0b) if (x0b + n_l) == ctb else cand_up
                cand_up_right = bool(sap and (x0 + n_l) < W)
                # lc->end_of_tiles_y = FFMIN(y_ctb + ctb_size, height): the bottom of the CURRENT CTB (hevc.c:2619)
                cand_bottom_left = False if (y0 + n_l) >= min(cty + ctb, H) else cand_left
                ops.append(dict(t="intra", x0=x0, y0=y0, log2=log2, c_idx=c_idx, mode=mode,
                                cands=[int(cand_bottom_left), int(cand_left), int(cand_up_left), int(cand_up), int(cand_up_right)]))

            cu(ctx_, cty, log2_ctb)

    # ---- in-loop filters
    fops = []
    for vertical in (1, 0):
        for c_idx in range(3):
            w, h = (W, H) if c_idx == 0 else (W // 2, H // 2)
            for y in range(0, h, 8):
                for x in range(0, w, 8):
                    if (vertical and x == 0) or (not vertical and y == 0) or rng.random() < 0.4:
                        continue
                    if x + 8 > w or y + 8 > h:
                        continue
                    fops.append(dict(t="dbk", c_idx=c_idx, x=x, y=y, vertical=vertical, beta=int(rng.integers(0, 65)),
                                     tc=[int(rng.integers(0, 14)), int(rng.integers(0, 14))],
                                     no_p=[int(rng.random() < 0.05), int(rng.random() < 0.05)],
                                     no_q=[int(rng.random() < 0.05), int(rng.random() < 0.05)]))
    for c_idx in range(3):
        w, h = (W, H) if c_idx == 0 else (W // 2, H // 2)
        cs = ctb if c_idx == 0 else ctb // 2
        for y in range(0, h, cs):
            for x in range(0, w, cs):
                if rng.random() < 0.35:
                    continue
                bw, bh = min(cs, w - x), min(cs, h - y)
                band = rng.random() < 0.3
                ov = [0] + [int(v) << (bd - 8 if bd <= 10 else 2) for v in rng.integers(-7, 8, size=4)]
                fops.append(dict(t="sao", c_idx=c_idx, x=x, y=y, w=bw, h=bh, band=band,
                                 klass=int(rng.integers(0, 32)) if band else int(rng.integers(0, 4)), offset_val=ov,
                                 borders=[int(x == 0), int(y == 0), int(x + bw == w), int(y + bh == h)]))
    return ops, fops
