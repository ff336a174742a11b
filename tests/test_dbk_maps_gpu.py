"""-m gpu parity test of ohevc_dev_deblock_maps: deblocking derived on the device from the decoder's maps.

Expected pictures: the loops of deblocking_filter_CTB (hevc_filter.c:345-581) restated here CTB by CTB in Python - they yield the
table calls the reference would make (plane, position, beta, tc[2], no_p[2], no_q[2]) - applied by the CPU oracle's filters.  The
kernel derives the same parameters per edge in closed form (which neighbour CTB's offsets an edge next to a CTB boundary gets is the
part worth pinning), so random per-CTB offsets, random boundary strengths, a random QP map and a random pcm map are used.
"""
import ctypes as C

import numpy as np
import pytest

from openhevc_amd import lib as L
import gpu_util as G
from test_filters_gpu import smooth_plane

pytestmark = pytest.mark.gpu

TC = [0] * 18 + [1] * 9 + [2] * 4 + [3] * 4 + [4] * 3 + [5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24]       # hevc_filter.c:50-54
BETA = [0] * 16 + list(range(6, 19)) + list(range(20, 66, 2))                                                           # :56-60
QPC = [29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37]                                                          # :65-67
assert len(TC) == 54 and len(BETA) == 52


def clip(v, lo, hi):
    return lo if v < lo else hi if v > hi else v


def reference_calls(m):
    """deblocking_filter_CTB for every CTB of the picture; returns [(vertical, plane, x, y, beta, tc0, tc1, no_p, no_q)]."""
    W, H, l2 = m["width"], m["height"], m["log2_ctb"]
    cfi = m["cfi"]
    hs, vs = int(cfi in (1, 2)), int(cfi == 1)
    h, v = 1 << hs, 1 << vs
    ctb, bw = 1 << l2, m["bs_width"]
    ctb_w = (W + ctb - 1) >> l2
    vbs, hbs = m["vertical_bs"], m["horizontal_bs"]
    qpy = lambda x, y: int(m["qp"][(x >> m["log2_min_cb"]) + (y >> m["log2_min_cb"]) * m["min_cb_width"]])      # get_qPy :144-150
    def pcm(x, y):                                                                                                # get_pcm :325-338
        if m["is_pcm"] is None:
            return 0
        if x < 0 or y < 0:
            return 2
        xp, yp = x >> m["log2_min_pu"], y >> m["log2_min_pu"]
        if xp >= m["min_pu_width"] or yp >= m["min_pu_height"]:
            return 2
        return int(m["is_pcm"][yp * m["min_pu_width"] + xp])
    tc_luma = lambda qp, bs, off: TC[clip(qp + 2 * (bs - 1) + ((off >> 1) << 1), 0, 53)]                          # TC_CALC :340-343
    def tc_chroma(qp_y, c, off):                                                                                  # chroma_tc :62-89
        qp_i = clip(qp_y + (m["cb_qp_offset"] if c == 1 else m["cr_qp_offset"]), 0, 57)
        if cfi == 1:
            qp = qp_i if qp_i < 30 else qp_i - 6 if qp_i > 43 else QPC[qp_i - 30]
        else:
            qp = clip(qp_i, 0, 51)
        return TC[clip(qp + 2 + off, 0, 53)]
    calls = []
    for y0 in range(0, H, ctb):
        for x0 in range(0, W, ctb):
            rs = (x0 >> l2) + (y0 >> l2) * ctb_w
            cur_beta, cur_tc = (int(t) for t in m["deblock"][rs])
            left_beta, left_tc = (int(t) for t in m["deblock"][rs - 1]) if x0 else (0, 0)
            x_end, y_end = min(x0 + ctb, W), min(y0 + ctb, H)
            tc_offset, beta_offset = cur_tc, cur_beta
            for y in range(y0, y_end, 8):                                         # vertical luma :385-420
                for x in range(x0 if x0 else 8, x_end, 8):
                    bs0, bs1 = int(vbs[(x + y * bw) >> 2]), int(vbs[(x + (y + 4) * bw) >> 2])
                    if not (bs0 or bs1):
                        continue
                    qp = (qpy(x - 1, y) + qpy(x, y) + 1) >> 1
                    calls.append((1, 0, x, y, BETA[clip(qp + beta_offset, 0, 51)], tc_luma(qp, bs0, tc_offset) if bs0 else 0, tc_luma(qp, bs1, tc_offset) if bs1 else 0,
                                  [pcm(x - 1, y), pcm(x - 1, y + 4)], [pcm(x, y), pcm(x, y + 4)]))
            if cfi:                                                               # vertical chroma :423-476
                for y in range(y0, y_end, 8 * v):
                    for x in range(x0 if x0 else 8 * h, x_end, 8 * h):
                        bs0, bs1 = int(vbs[(x + y * bw) >> 2]), int(vbs[(x + (y + 4 * v) * bw) >> 2])
                        if not (bs0 == 2 or bs1 == 2):
                            continue
                        qp0 = (qpy(x - 1, y) + qpy(x, y) + 1) >> 1
                        # (the reference evaluates qp1 even where bs1 != 2 and y + 4v is the first row below the picture; unused there)
                        qp1 = (qpy(x - 1, y + 4 * v) + qpy(x, y + 4 * v) + 1) >> 1 if bs1 == 2 else 0
                        for c in (1, 2):
                            calls.append((1, c, x >> hs, y >> vs, 0, tc_chroma(qp0, c, tc_offset) if bs0 == 2 else 0, tc_chroma(qp1, c, tc_offset) if bs1 == 2 else 0,
                                          [pcm(x - 1, y), pcm(x - 1, y + 4 * v)], [pcm(x, y), pcm(x, y + 4 * v)]))
            x_end2 = x_end                                                        # horizontal luma :479-519
            if x_end != W:
                x_end -= 8
            for y in range(y0 if y0 else 8, y_end, 8):
                beta_offset = left_beta if x0 else cur_beta
                for x in range(x0 - 8 if x0 else 0, x_end, 8):
                    bs0, bs1 = int(hbs[(x + y * bw) >> 2]), int(hbs[((x + 4) + y * bw) >> 2])
                    if bs0 or bs1:
                        qp = (qpy(x, y - 1) + qpy(x, y) + 1) >> 1
                        calls.append((0, 0, x, y, BETA[clip(qp + beta_offset, 0, 51)], tc_luma(qp, bs0, tc_offset) if bs0 else 0, tc_luma(qp, bs1, tc_offset) if bs1 else 0,
                                      [pcm(x, y - 1), pcm(x + 4, y - 1)], [pcm(x, y), pcm(x + 4, y)]))
                    beta_offset = cur_beta
            if cfi:                                                               # horizontal chroma :522-579
                if x_end2 != W:
                    x_end = x_end2 - 8 * h
                for y in range(y0 if y0 else 8 * v, y_end, 8 * v):
                    tc_offset = left_tc if x0 else cur_tc
                    for x in range(x0 - 8 * h if x0 else 0, x_end, 8 * h):
                        bs0, bs1 = int(hbs[(x + y * bw) >> 2]), int(hbs[((x + 4 * h) + y * bw) >> 2])
                        if bs0 == 2 or bs1 == 2:
                            qp0 = (qpy(x, y - 1) + qpy(x, y) + 1) >> 1 if bs0 == 2 else 0
                            qp1 = (qpy(x + 4 * h, y - 1) + qpy(x + 4 * h, y) + 1) >> 1 if bs1 == 2 else 0
                            for c in (1, 2):
                                calls.append((0, c, x >> hs, y >> vs, 0, tc_chroma(qp0, c, tc_offset) if bs0 == 2 else 0, tc_chroma(qp1, c, cur_tc) if bs1 == 2 else 0,
                                              [pcm(x, y - 1), pcm(x + 4 * h, y - 1)], [pcm(x, y), pcm(x + 4 * h, y)]))
                        tc_offset = cur_tc
    return calls


# form: 0 = a lane per 4-line luma segment, packed 16-bit arithmetic (what pictures up to 10 bit take), 1 = a lane per line (deeper
# pictures, and every chroma edge)
@pytest.mark.parametrize("form", [0, 1])
@pytest.mark.parametrize("bd,cfi,log2_ctb,W,H,with_pcm", [(8, 1, 4, 208, 120, 0), (8, 1, 6, 416, 240, 1), (10, 1, 5, 200, 136, 1), (8, 2, 5, 208, 120, 0),
                                                          (10, 3, 4, 136, 72, 1), (14, 1, 6, 192, 136, 0), (8, 1, 6, 64, 64, 0), (9, 0, 4, 80, 48, 1),
                                                          (8, 1, 5, 1032, 40, 1)])
def test_deblocking_derived_on_the_device(oracle, bd, cfi, log2_ctb, W, H, with_pcm, form):
    rng = np.random.default_rng(900 + bd + 10 * cfi + log2_ctb + W)
    hs, vs = int(cfi in (1, 2)), int(cfi == 1)
    planes = [smooth_plane(rng, bd, H, W), smooth_plane(rng, bd, H >> vs, W >> hs), smooth_plane(rng, bd, H >> vs, W >> hs)]
    bw, bh = W >> 2, H >> 2
    ctb = 1 << log2_ctb
    ctb_w, ctb_h = (W + ctb - 1) >> log2_ctb, (H + ctb - 1) >> log2_ctb
    m = dict(width=W, height=H, log2_ctb=log2_ctb, cfi=cfi, bs_width=bw, log2_min_cb=3, min_cb_width=W >> 3, log2_min_pu=2, min_pu_width=W >> 2, min_pu_height=H >> 2,
             cb_qp_offset=int(rng.integers(-6, 7)), cr_qp_offset=int(rng.integers(-6, 7)))
    # sizes as hevc.c:170-171 allocates them; the rows past the picture stay 0 like in the reference (av_mallocz, never written)
    vb = np.zeros(bw * (bh + (4 << vs)), np.uint8)
    hb = np.zeros((bw + (4 << hs)) * bh, np.uint8)
    vb[:bw * bh] = rng.choice([0, 0, 1, 2], size=bw * bh)
    hb[:bw * bh] = rng.choice([0, 0, 1, 2], size=bw * bh)
    m["vertical_bs"], m["horizontal_bs"] = vb, hb
    m["qp"] = rng.integers(8, 52, size=(W >> 3) * (H >> 3)).astype(np.int8)
    m["deblock"] = rng.integers(-6, 7, size=(ctb_w * ctb_h, 2)).astype(np.int8) * 2            # slice_{beta,tc}_offset_div2 * 2
    m["deblock"][:, 1] += rng.integers(0, 2, size=ctb_w * ctb_h).astype(np.int8)               # odd tc offsets: TC_CALC's >> 1 << 1
    m["is_pcm"] = (rng.random((W >> 2) * (H >> 2)) < 0.1).astype(np.uint8) if with_pcm else None

    want = [p.copy() for p in planes]
    calls = reference_calls(m)
    for vertical in (1, 0):                       # the executor's order: all vertical edges, then all horizontal ones
        for (vert, pl, x, y, beta, tc0, tc1, no_p, no_q) in calls:
            if vert != vertical:
                continue
            np_, nq_ = [int(bool(t)) for t in no_p], [int(bool(t)) for t in no_q]
            if pl == 0:
                oracle.deblock_luma(bd, vertical, want[0], x, y, beta, [tc0, tc1], np_, nq_)
            else:
                oracle.deblock_chroma(bd, vertical, want[pl], x, y, [tc0, tc1], np_, nq_)
    assert len(calls) > 50

    d = [G.to_dev(p) for p in planes]
    keep = [G.to_dev(a) for a in (vb, hb, m["qp"], m["deblock"])] + ([G.to_dev(m["is_pcm"])] if with_pcm else [])
    dm = L.DbkMaps(vertical_bs=keep[0].data_ptr(), horizontal_bs=keep[1].data_ptr(), qp_y_tab=keep[2].data_ptr(), deblock=keep[3].data_ptr(),
                   is_pcm=keep[4].data_ptr() if with_pcm else None, bs_width=bw, min_cb_width=W >> 3, deblock_stride=2, min_pu_width=W >> 2, min_pu_height=H >> 2,
                   width=W, height=H, log2_ctb_size=log2_ctb, log2_min_cb_size=3, log2_min_pu_size=2, chroma_format_idc=cfi,
                   cb_qp_offset=m["cb_qp_offset"], cr_qp_offset=m["cr_qp_offset"])
    lib = L.load_library()
    lib.ohevc_debug_deblock_segment_launches.restype = C.c_longlong
    before = lib.ohevc_debug_deblock_segment_launches()
    previous = lib.ohevc_debug_set_deblock_variant(form)
    try:
        for vertical in (1, 0):
            L.dev_deblock_maps(G.planes3(d), bd, dm, vertical, G.stream())
        G.sync()
    finally:
        lib.ohevc_debug_set_deblock_variant(previous)
    assert lib.ohevc_debug_deblock_segment_launches() - before == (2 if form == 0 and bd <= 10 else 0), "not the form this case is about"
    changed = 0
    for pl in range(3):
        got = G.to_host(d[pl], planes[pl].dtype)
        bad = np.argwhere(got != want[pl])
        assert bad.size == 0, f"plane {pl}: {len(bad)} mismatches, first {bad[:4].tolist()}"
        changed += int((got != planes[pl]).sum())
    assert changed > 100, "test content did not trigger the filters"
