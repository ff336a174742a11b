"""-m gpu parity of the device-side boundary strengths (second half): ohevc_dev_boundary_strengths over a motion field, and over
the motion field ohevc_dev_motion_grid rebuilds from a picture's luma motion-compensation jobs, against oracle/hevc_oracle.c's
ohor_boundary_strengths - which tests/test_oracle_vs_reference.py pins to the arrays the reference's own function fills
(hevc_filter.c:584-700, 805-941).

Inputs: random coding / prediction / transform trees (every PartMode, AMP, 8x4 / 4x8 blocks, motion drawn so that neighbours are often equal,
a quarter sample apart, a whole sample apart, or swap their two references), random slice / tile flags per CTB; and - where the reference
build is at hand - the pictures of committed streams as the reference's front end left them (oracle/null_hooks.c's tap)."""
import os

import numpy as np
import pytest

from openhevc_amd import lib as L
from oracle import pyoracle as po
import gpu_util as G

pytestmark = pytest.mark.gpu
ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so")


def synth_picture(rng, W, H, log2_ctb, l2pu, n_refs=3):
    """-> field (po.BS_FIELD per min PU, poc = reference slot), cbf_luma (per 4x4), calls (L.BS_CALL), pus [(x, y, w, h, pred_flag, mv0, mv1, ref0, ref1)]"""
    pu_w, pu_h, tb_w, tb_h = W >> l2pu, H >> l2pu, W >> 2, H >> 2
    field = np.zeros(pu_w * pu_h, po.BS_FIELD)
    cbf = np.zeros(tb_w * tb_h, np.uint8)
    calls, pus = [], []
    min_cb = 2 << l2pu
    base = [(int(rng.integers(-64, 64)), int(rng.integers(-64, 64))) for _ in range(4)]
    ctb = 1 << log2_ctb
    ctb_flags = {}

    def motion():
        b = base[int(rng.integers(0, len(base)))]
        d = [0, 0, 1, 3, 4, -4, 5, -3]
        return (b[0] + d[int(rng.integers(0, 8))], b[1] + d[int(rng.integers(0, 8))])

    def put_pu(x, y, w, h):
        small = w + h <= 12
        pf = int(rng.choice([1, 2] if small else [1, 2, 3, 3]))                      # no bi-prediction of 8x4 / 4x8 (hevc.c:1902-1907)
        mv0, mv1, r0, r1 = motion(), motion(), int(rng.integers(0, n_refs)), int(rng.integers(0, n_refs))
        if pf == 3 and rng.random() < 0.3:
            r1 = r0                                                                  # both references one picture: the first branch of boundary_strength
        for yy in range(y >> l2pu, (y + h) >> l2pu):
            for xx in range(x >> l2pu, (x + w) >> l2pu):
                e = field[yy * pu_w + xx]
                e["pred_flag"] = pf
                if pf & 1:
                    e["mv"][0] = mv0; e["poc"][0] = r0
                if pf & 2:
                    e["mv"][1] = mv1; e["poc"][1] = r1
        pus.append((x, y, w, h, pf, mv0, mv1, r0, r1))

    def tu_tree(x, y, log2, fl):
        if x >= W or y >= H:
            return
        if log2 > 2 and (log2 > 5 or rng.random() < 0.45):
            n = 1 << (log2 - 1)
            for (dx, dy) in ((0, 0), (n, 0), (0, n), (n, n)):
                tu_tree(x + dx, y + dy, log2 - 1, fl)
            return
        n = 1 << log2
        if rng.random() < 0.35:
            for yy in range(y >> 2, (y + n) >> 2):
                cbf[yy * tb_w + (x >> 2):yy * tb_w + ((x + n) >> 2)] = 1
        calls.append((x, y, log2, fl))

    def cu_tree(x, y, log2):
        if x >= W or y >= H:
            return
        n = 1 << log2
        if n > min_cb and (x + n > W or y + n > H or rng.random() < 0.55):
            for (dx, dy) in ((0, 0), (n >> 1, 0), (0, n >> 1), (n >> 1, n >> 1)):
                cu_tree(x + dx, y + dy, log2 - 1)
            return
        fl = ctb_flags.setdefault((x >> log2_ctb, y >> log2_ctb), int(rng.choice([0, 0, 16, 16, 1 | 4, 2 | 8, 1 | 4 | 16, 3 | 12, 1, 8 | 16])))
        if rng.random() < 0.2:                                                       # intra: the field stays PF_INTRA
            tu_tree(x, y, log2, fl)
            return
        q = n >> 2
        modes = ["2Nx2N", "2NxN", "Nx2N"] + (["2NxnU", "2NxnD", "nLx2N", "nRx2N"] if n > min_cb else [])
        mode = modes[int(rng.integers(0, len(modes)))]
        parts = {"2Nx2N": [(0, 0, n, n)], "2NxN": [(0, 0, n, n >> 1), (0, n >> 1, n, n >> 1)], "Nx2N": [(0, 0, n >> 1, n), (n >> 1, 0, n >> 1, n)],
                 "2NxnU": [(0, 0, n, q), (0, q, n, n - q)], "2NxnD": [(0, 0, n, n - q), (0, n - q, n, q)],
                 "nLx2N": [(0, 0, q, n), (q, 0, n - q, n)], "nRx2N": [(0, 0, n - q, n), (n - q, 0, q, n)]}[mode]
        for (dx, dy, w, h) in parts:
            put_pu(x + dx, y + dy, w, h)
        if rng.random() < 0.3:
            calls.append((x, y, log2, fl))                                           # skipped CU: one call for the coding block (hevc.c:2400)
        else:
            tu_tree(x, y, log2, fl)

    for y in range(0, H, ctb):
        for x in range(0, W, ctb):
            cu_tree(x, y, log2_ctb)
    c = np.zeros(len(calls), L.BS_CALL)
    for i, (x, y, l2, fl) in enumerate(calls):
        c[i] = (x, y, l2, fl, 0)
    return field, cbf, c, pus


def jobs_of_pus(rng, pus):
    """the tiles ohevc_rec_mc cuts (at most 16x16), luma; chroma jobs of the same blocks mixed in (the grid must ignore them)"""
    out = []
    for (x, y, w, h, pf, mv0, mv1, r0, r1) in pus:
        first, second = ((mv0, r0), (mv1, r1)) if pf == 3 else ((mv0, r0), None) if pf == 1 else ((mv1, r1), None)
        for ty in range(0, h, 16):
            for tx in range(0, w, 16):
                j = np.zeros(1, L.MC_JOB)[0]
                j["x"], j["y"], j["w"], j["h"] = x + tx, y + ty, min(16, w - tx), min(16, h - ty)
                j["sx0"], j["sy0"], j["mx0"], j["my0"], j["ref0"] = x + tx + (first[0][0] >> 2), y + ty + (first[0][1] >> 2), first[0][0] & 3, first[0][1] & 3, first[1]
                if second:
                    j["flags"] = L.MC_BI
                    j["sx1"], j["sy1"], j["mx1"], j["my1"], j["ref1"] = x + tx + (second[0][0] >> 2), y + ty + (second[0][1] >> 2), second[0][0] & 3, second[0][1] & 3, second[1]
                out.append(j)
                if rng.random() < 0.3:
                    k = j.copy()
                    k["plane"] = int(rng.integers(1, 3)); k["x"] >>= 1; k["y"] >>= 1; k["w"] = max(2, int(k["w"]) >> 1); k["h"] = max(2, int(k["h"]) >> 1)
                    k["sx0"] = 7; k["mx0"] = 5; k["ref0"] = 9
                    out.append(k)
    a = np.array(out, L.MC_JOB)
    return a[rng.permutation(len(a))]


def device_bs(geom, d_field, d_cbf, calls, n_bs, pu_w, pu_h, tb_w, tb_h, W, H):
    maps = L.BsMaps(mvf=d_field.data_ptr(), mvf_stride=20, off_mv=0, off_poc=8, off_pred_flag=16, pred_flag_bytes=4, cbf_luma=d_cbf.data_ptr(),
                    min_pu_width=pu_w, min_pu_height=pu_h, log2_min_pu_size=geom["log2_min_pu_size"], min_tb_width=tb_w, min_tb_height=tb_h,
                    log2_min_tb_size=geom["log2_min_tb_size"], log2_ctb_size=geom["log2_ctb_size"], bs_width=geom["bs_width"], width=W, height=H,
                    loop_filter_across_tiles=geom["loop_filter_across_tiles"])
    d_calls = G.to_dev(calls)
    d_v, d_h = G.zeros_dev(n_bs, np.uint8), G.zeros_dev(n_bs, np.uint8)
    L.dev_boundary_strengths(maps, d_calls.data_ptr(), len(calls), d_v.data_ptr(), d_h.data_ptr(), G.stream())
    G.sync()
    return G.to_host(d_v, np.uint8).copy(), G.to_host(d_h, np.uint8).copy()


def grid_of_jobs(jobs, pu_w, pu_h, l2pu, split=None):
    """split = k: the jobs as two arrays (the first k, the rest) through the one-launch-for-both entry point, as the ctx layer calls it"""
    d_grid = G.zeros_dev(pu_w * pu_h * L.MOTION_GRID_ENTRY, np.uint8)
    if split is None:
        d_jobs = G.to_dev(jobs)
        L.dev_motion_grid(d_jobs.data_ptr(), len(jobs), d_grid.data_ptr(), pu_w, pu_h, l2pu, G.stream())
    else:
        d_a, d_b = G.to_dev(jobs[:max(split, 1)]), G.to_dev(jobs[split:] if split < len(jobs) else jobs[:1])
        L.dev_motion_grid2(d_a.data_ptr() if split else 0, split, d_b.data_ptr() if split < len(jobs) else 0, len(jobs) - split, d_grid.data_ptr(), pu_w, pu_h, l2pu,
                           G.stream())
    G.sync()
    return d_grid


def compare(what, got, want):
    for g, w, name in zip(got, want, ("vertical", "horizontal")):
        bad = np.flatnonzero(g != w)
        assert bad.size == 0, f"{what} {name}_bs: {bad.size} entries differ, first {bad[:4].tolist()} device {g[bad[:4]].tolist()} oracle {w[bad[:4]].tolist()}"


@pytest.mark.parametrize("W,H,log2_ctb,l2pu", [(416, 240, 6, 2), (200, 136, 5, 2), (136, 72, 4, 2), (256, 192, 6, 3), (64, 64, 6, 2), (1920, 1080 // 8 * 8, 6, 2)])
def test_boundary_strengths_from_field_and_from_mc_jobs(W, H, log2_ctb, l2pu):
    rng = np.random.default_rng(500 + W + H + log2_ctb + l2pu)
    for across_tiles in (1, 0):
        field, cbf, calls, pus = synth_picture(rng, W, H, log2_ctb, l2pu)
        pu_w, pu_h, tb_w, tb_h = W >> l2pu, H >> l2pu, W >> 2, H >> 2
        geom = dict(min_pu_width=pu_w, log2_min_pu_size=l2pu, min_tb_width=tb_w, log2_min_tb_size=2, log2_ctb_size=log2_ctb, bs_width=W >> 2,
                    loop_filter_across_tiles=across_tiles)
        n_bs = (W >> 2) * (H >> 2)
        want = po.boundary_strengths(ORACLE, geom, field, cbf, calls.view(po.BS_CALL), n_bs)
        assert W * H < 20000 or (np.count_nonzero(want[0] == 1) > 10 and np.count_nonzero(want[0] == 2) > 10 and np.count_nonzero(want[1] == 0) > 10)
        d_cbf = G.to_dev(cbf)
        compare(f"{W}x{H} field", device_bs(geom, G.to_dev(field), d_cbf, calls, n_bs, pu_w, pu_h, tb_w, tb_h, W, H), want)
        jobs = jobs_of_pus(rng, pus)
        d_grid = grid_of_jobs(jobs, pu_w, pu_h, l2pu, split=None if across_tiles else int(rng.choice([0, len(jobs) // 3, len(jobs)])))
        grid = G.to_host(d_grid, np.uint8).view(po.BS_FIELD)
        assert np.array_equal(grid["pred_flag"] != 0, field["pred_flag"] != 0)           # exactly the inter-predicted units were written
        compare(f"{W}x{H} MC jobs", device_bs(geom, d_grid, d_cbf, calls, n_bs, pu_w, pu_h, tb_w, tb_h, W, H), want)


@pytest.mark.parametrize("name", ["ra_8b_ctb64", "ldb_8b", "tiles_nolf", "slices_nolf", "weighted", "ra_8b_foll_leaf", "bqmall_geometry_dense_qp22"])
def test_boundary_strengths_of_committed_streams(name):
    """the reference's own arrays (not only the restatement) on real syntax: position, size and flags of every call, the motion field and
    the cbf_luma map as the reference's front end left them; the MC-job path gets one 4x4 ... min-PU-sized job per inter-predicted unit"""
    from oracle import pystream as ps
    from test_stream_cpu import load_golden
    if not ps.have("null"):
        pytest.skip("oracle/_ref/libopenhevc_null.so not built")
    aus, _ = load_golden(name)
    pictures = 0
    with ps.Decoder("null", 1, 1) as d:
        for i, au in enumerate(aus):
            po.bs_tap(d.L, True)
            assert d.L.ohdec_decode(d.h, au, len(au), i + 1) >= 0
            t = po.bs_tap_fetch(d.L)
            if t is None:
                continue
            g, W, H = t["geom"], t["width"], t["height"]
            pu_w, pu_h, tb_w, tb_h, l2pu = g["min_pu_width"], t["min_pu_height"], g["min_tb_width"], t["min_tb_height"], g["log2_min_pu_size"]
            want = (t["vertical_bs"], t["horizontal_bs"])
            calls = t["calls"].view(L.BS_CALL)
            d_cbf = G.to_dev(t["cbf_luma"])
            compare(f"{name} picture {i} field", device_bs(g, G.to_dev(t["mvf"]), d_cbf, calls, t["n_bs"], pu_w, pu_h, tb_w, tb_h, W, H), want)
            # the same motion as MC jobs: POCs -> reference slots
            f = t["mvf"]
            used = [f["poc"][:, l][(f["pred_flag"] >> l) & 1 == 1] for l in (0, 1)]            # (entries of unused lists are whatever the buffer held)
            slots = {int(p): k for k, p in enumerate(np.unique(np.concatenate(used)))}
            u = 1 << l2pu
            jobs = []
            for k in np.flatnonzero(f["pred_flag"]):
                e, x, y = f[k], (int(k) % pu_w) * u, (int(k) // pu_w) * u
                lists = [l for l in (0, 1) if int(e["pred_flag"]) & (1 << l)]
                j = np.zeros(1, L.MC_JOB)[0]
                j["x"], j["y"], j["w"], j["h"] = x, y, u, u
                a = lists[0]
                j["sx0"], j["sy0"], j["mx0"], j["my0"], j["ref0"] = x + (int(e["mv"][a][0]) >> 2), y + (int(e["mv"][a][1]) >> 2), int(e["mv"][a][0]) & 3, int(e["mv"][a][1]) & 3, slots[int(e["poc"][a])]
                if len(lists) == 2:
                    j["flags"] = L.MC_BI
                    j["sx1"], j["sy1"], j["mx1"], j["my1"], j["ref1"] = x + (int(e["mv"][1][0]) >> 2), y + (int(e["mv"][1][1]) >> 2), int(e["mv"][1][0]) & 3, int(e["mv"][1][1]) & 3, slots[int(e["poc"][1])]
                jobs.append(j)
            if jobs:
                d_grid = grid_of_jobs(np.array(jobs, L.MC_JOB), pu_w, pu_h, l2pu)
            else:
                d_grid = G.zeros_dev(pu_w * pu_h * L.MOTION_GRID_ENTRY, np.uint8)
            compare(f"{name} picture {i} MC jobs", device_bs(g, d_grid, d_cbf, calls, t["n_bs"], pu_w, pu_h, tb_w, tb_h, W, H), want)
            pictures += 1
        po.bs_tap(d.L, False)
    assert pictures > 0
