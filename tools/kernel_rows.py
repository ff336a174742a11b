"""Per-kernel rows of bench.py's `kernels` object: every kernel family of the hot path beside the graded one, out of HBM-resident rings,
with a sampled bit-exact check against the CPU oracle (oracle/liboracle.so; test infrastructure, called only as the checker).

Workloads are those of tools/bench_kernels.py --resident --planes 8 (the rows of profiles/r0*_bench_kernels_*.jsonl): a 3840 x 17280 luma
plane (eight 4K pictures stacked) tiled with jobs of one kind; every launch takes the next entry of a ring of pictures that is at least
RING_BYTES large (destinations, and for motion compensation / SAO their sources too), written once at set-up, so nothing a launch touches is
cache-resident from the launch before: the GB/s are HBM numbers.  Time = median over the launches of HIP events recorded on the launch
stream (torch's current stream: the launches are issued on it), around bursts of eight launches.  `achieved` = algorithmic bytes per launch (SURVEY.md 8d per-unit figures
x units) / that time; `frac` = achieved / 8 TB/s.

Check: one more launch on a fresh picture whose state before the launch was kept; a sample of the launch's units (blocks, edges, CTBs) is
recomputed by the oracle from that state and compared sample by sample with what the device wrote."""
import os
import sys

import ctypes as C

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openhevc_amd import lib as L  # noqa: E402

W, H = 3840, 2160 * 8
PEAK = 8000.0
RING_BYTES = 1 << 30
N_CHECK = 48

# H.265 table 8-12 (beta', tc' by Q), as the reference holds them (hevc_filter.c:50-60,62-89)
TC_TABLE = [0] * 18 + [1] * 9 + [2] * 4 + [3] * 4 + [4] * 3 + [5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24]
BETA_TABLE = [0] * 16 + list(range(6, 19)) + list(range(20, 66, 2))


def _dev(a):
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.uint8) if a.dtype.fields is not None else a).cuda()


def _smooth_pic(bd, g, luma_only=False):
    """a diagonal ramp + small noise: deblocking really filters and SAO classifies real edges (white noise defeats both decisions)"""
    dt = torch.uint8 if bd == 8 else torch.int16

    def mk(h, w):
        yy = torch.arange(h, device="cuda", dtype=torch.int32)[:, None]
        xx = torch.arange(w, device="cuda", dtype=torch.int32)[None, :]
        base = ((xx + yy) >> 3) % ((1 << bd) - 16) + 8
        return (base + torch.randint(-3, 4, (h, w), dtype=torch.int32, device="cuda", generator=g)).to(dt)
    return [mk(H, W), None, None] if luma_only else [mk(H, W), mk(H // 2, W // 2), mk(H // 2, W // 2)]


def _bytes(pic):
    return sum(t.numel() * t.element_size() for t in pic if t is not None)


def _time(launch, fresh, extra=None, extra_bytes=0, reps=8, burst=8):
    """launch(pic, ex); ring of fresh() pictures (+ extra(k) sources) of at least RING_BYTES.  One measurement = HIP events around a BURST of
    launches on consecutive ring entries, divided by their number (a lone 50-150 us kernel between two events reads 10-15 us long: the events'
    own latency; rocprofv3's per-dispatch durations - profiles/r4p_* - are what the burst form agrees with); median over reps bursts after
    one warm-up burst."""
    st = torch.cuda.current_stream()
    first = fresh()
    n_ring = max(2, -(-RING_BYTES // max(1, _bytes(first) + extra_bytes)))
    ring = [first] + [fresh() for _ in range(n_ring - 1)]
    ex = [extra(k) for k in range(n_ring)] if extra else [None] * n_ring
    burst = max(1, burst if n_ring >= 3 else n_ring)       # (the ring is passed over more than once per burst: an entry is 100-300 MB, the caches hold none of it by then)
    torch.cuda.synchronize()
    ts = []
    k = 0
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(burst):
            launch(ring[k % n_ring], ex[k % n_ring])
            k += 1
        b.record(st)
        torch.cuda.synchronize()
        if r >= 1:
            ts.append(a.elapsed_time(b) / burst)
    ring_bytes = n_ring * (_bytes(first) + extra_bytes)
    del ring, ex
    torch.cuda.empty_cache()
    return float(np.median(ts)), ring_bytes


def _row(ms, ring_bytes, alg_bytes, pixels, bad, n_checked, what):
    gbs = alg_bytes / ms / 1e6
    return {"kernel_ms": round(ms, 4), "achieved": round(gbs, 1), "unit": "GB/s", "frac": round(gbs / PEAK, 4), "algorithmic_bytes_per_launch": int(alg_bytes),
            "mpixel_per_s": round(pixels / ms / 1e3, 1), "ring_bytes": int(ring_bytes), "checked": bool(n_checked and bad == 0),
            "check": {"units": n_checked, "mismatches": bad}, "workload": what}


def _np(t, bd):
    a = t.cpu().numpy()
    return a.view(np.uint16) if bd > 8 else a


def _window(ref, sx, sy, w, h):
    rows = np.clip(np.arange(sy - 3, sy + h + 4), 0, ref.shape[0] - 1)
    cols = np.clip(np.arange(sx - 3, sx + w + 4), 0, ref.shape[1] - 1)
    return np.ascontiguousarray(ref[rows][:, cols])


def mc_rows(bd, orc, po, g, rng, st, out):
    P = 2 if bd > 8 else 1
    for bw, bh, bi, small in ((8, 8, 0, True), (16, 16, 1, False)):
        xs, ys = np.meshgrid(np.arange(0, W - bw + 1, bw), np.arange(0, H - bh + 1, bh))
        n = xs.size
        j = np.zeros(n, L.MC_JOB)
        j["x"], j["y"], j["w"], j["h"], j["plane"] = xs.ravel(), ys.ravel(), bw, bh, 0
        j["flags"] = L.MC_BI if bi else 0
        for s in ("0", "1"):
            j["sx" + s] = j["x"].astype(np.int32) + rng.integers(-16, 17, n)
            j["sy" + s] = j["y"].astype(np.int32) + rng.integers(-16, 17, n)
            j["mx" + s], j["my" + s] = rng.integers(0, 4, n), rng.integers(0, 4, n)
        j["ref1"] = 1
        d_jobs = _dev(j)

        def sources(k):
            rr = [_smooth_pic(bd, g) for _ in range(2)]
            return rr, _dev(L.planes_table(rr))

        def launch(pic, ex):
            if small:
                L.dev_mc_batch_small(L.planes_of(pic), ex[1].data_ptr(), 2, bd, d_jobs.data_ptr(), n, st())
            else:
                L.dev_mc_batch_bounded(L.planes_of(pic), ex[1].data_ptr(), 2, bd, d_jobs.data_ptr(), n, bw, bh, st())
        src_bytes = 2 * _bytes(_smooth_pic(bd, g))
        ms, ring = _time(launch, lambda: _smooth_pic(bd, g), sources, src_bytes)
        # check: sampled blocks recomputed from the reference pictures (hevcdsp_template.c:731-983: plain h/v/hv pass, uni / bi rounding)
        pic, ex = _smooth_pic(bd, g), sources(0)
        launch(pic, ex)
        torch.cuda.synchronize()
        got, r0, r1 = _np(pic[0], bd), _np(ex[0][0][0], bd), _np(ex[0][1][0], bd)
        bad = 0
        for k in rng.integers(0, n, N_CHECK):
            q = j[k]
            x, y = int(q["x"]), int(q["y"])
            if not bi:
                want = orc.mc(bd, True, po.MC_UNI, _window(r0, int(q["sx0"]), int(q["sy0"]), bw, bh), 3, 3, bw, bh, int(q["mx0"]), int(q["my0"]))
            else:
                tmp = orc.mc(bd, True, po.MC_PUT, _window(r0, int(q["sx0"]), int(q["sy0"]), bw, bh), 3, 3, bw, bh, int(q["mx0"]), int(q["my0"]))
                src2 = np.zeros((bh, 64), np.int16)
                src2[:, :bw] = tmp
                want = orc.mc(bd, True, po.MC_BI, _window(r1, int(q["sx1"]), int(q["sy1"]), bw, bh), 3, 3, bw, bh, int(q["mx1"]), int(q["my1"]), src2=src2)
            bad += not np.array_equal(got[y:y + bh, x:x + bw], want)
        alg = n * ((1 + bi) * P * (bw + 7) * (bh + 7) + P * bw * bh)
        out[f"mc_luma_{bw}x{bh}_{'bi' if bi else 'uni'}_{bd}bit"] = _row(
            ms, ring, alg, n * bw * bh, bad, N_CHECK,
            f"{n} luma blocks {bw}x{bh}, {'bi' if bi else 'uni'}-prediction, random quarter-sample phases and vectors within +-16 samples, "
            f"{'small-block entry point (four blocks per matrix-core tile)' if small else 'tile entry point'}; put_hevc_qpel_{'bi' if bi else 'uni'}_* (hevcdsp_template.c:731-983)")


def intra_rows(bd, orc, po, g, rng, st, out):
    P = 2 if bd > 8 else 1
    for log2 in (2, 5):
        nn = 1 << log2
        # blocks on a sparse grid (every other block position) inside the picture's interior coding-tree blocks.  Which neighbour groups a block may
        # use follows from the z-scan order of its position inside the 64x64 CTB (hevcpred_template.c:105-114) - the job builder of the C ABI
        # derives it, once per position of one interior CTB; the pattern repeats from CTB to CTB
        xs, ys = np.meshgrid(np.arange(64 + nn, W - 64 - nn, 2 * nn), np.arange(64 + nn, H - 64 - nn, 2 * nn))
        n = xs.size
        geom = L.IntraGeom(W, H, 1, 6, 2, 1, 0, 0)
        pattern = {}
        for py in range(nn % 64, 64, 2 * nn):
            for px_ in range(nn % 64, 64, 2 * nn):
                pattern[(px_, py)] = L.intra_make_job(geom, 64 + px_, 64 + py, log2, 0, 0, [1, 1, 1, 1, 1])[0]
        j = np.zeros(n, L.INTRA_JOB)
        fx, fy = xs.ravel() % 64, ys.ravel() % 64
        for (px_, py), jb in pattern.items():
            m = (fx == px_) & (fy == py)
            for f in ("flags", "bottom_left_size", "top_right_size", "flags2", "log2_ctb_size"):
                j[f][m] = jb[f]
        j["x"], j["y"], j["log2_size"], j["mode"] = xs.ravel(), ys.ravel(), log2, rng.integers(0, 35, n)
        r = np.zeros(n, L.TU_JOB)
        r["x"], r["y"], r["reserved0"], r["coeff_off"] = j["x"], j["y"], L.TU_IDCT + 1, np.arange(n, dtype=np.uint32) * nn * nn
        # the order the ctx layer stages a dependency level in (ohevc_intra_sort_level, include/ohevc_hip.h): by size, otherwise as recorded
        cnt = (C.c_int32 * 4)()
        L.check(L.load_library().ohevc_intra_sort_level(j.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), C.c_int(n), cnt))
        counts = list(cnt)
        assert counts[log2 - 2] == n
        d_jobs = _dev(j)
        d_res = _dev(r)
        cf = torch.randint(-256, 256, (n * nn * nn,), dtype=torch.int16, device="cuda", generator=g)

        def launch(pic, ex):
            L.dev_intra_recon_sorted(L.planes_of(pic), bd, d_jobs.data_ptr(), d_res.data_ptr(), counts, cf.data_ptr(), st())
        ms, ring = _time(launch, lambda: _smooth_pic(bd, g))
        pic = _smooth_pic(bd, g)
        before = [_np(t, bd).copy() for t in pic]
        launch(pic, None)
        torch.cuda.synchronize()
        got = _np(pic[0], bd)
        bad = 0
        for k in rng.integers(0, n, N_CHECK):
            x, y, mode = int(j["x"][k]), int(j["y"][k]), int(j["mode"][k])
            # (blocks sit on a sparse grid: a block's neighbours are untouched samples, the oracle may work on the picture as it was)
            orc.intra_pred(bd, before, W, H, x, y, log2, 0, mode, (1, 1, 1, 1, 1), chroma_format_idc=1, strong=1, smoothing_disabled=0,
                           log2_ctb_size=6, log2_min_tb_size=2)
            co = int(r["coeff_off"][k])                     # (the level sort permuted jobs and residual records together)
            orc.tu_batch(bd, po.TU_IDCT, log2, cf[co:co + nn * nn].cpu().numpy().reshape(1, nn, nn), before[0], np.array([[x, y]], np.int32))
            bad += not np.array_equal(got[y:y + nn, x:x + nn], before[0][y:y + nn, x:x + nn])
        alg = n * (P * (4 * nn + 1) + P * nn * nn + 2 * nn * nn)
        out[f"intra_{nn}x{nn}_with_residual_{bd}bit"] = _row(
            ms, ring, alg, n * nn * nn, bad, N_CHECK,
            f"{n} independent luma blocks {nn}x{nn}, all 35 modes (uniform), staged like a level of the ctx layer; prediction + the block's "
            f"inverse-DCT residual in one pass (packed kernel, N lanes per block); intra_pred / pred_planar / pred_dc / pred_angular (hevcpred_template.c:30-537) + idct + transform_add")
        del d_res, cf


def deblock_rows(bd, orc, po, g, rng, st, out):
    P = 2 if bd > 8 else 1
    bw, bh = W >> 2, H >> 2
    vb = np.zeros(bw * (bh + 8), np.uint8)
    hb = np.zeros((bw + 8) * bh, np.uint8)
    grid = np.zeros((bh, bw), np.uint8)
    grid[:, ::2] = 1                                  # vertical edges at x % 8 == 0 (bs index = x / 4)
    vb[:bw * bh] = grid.ravel()
    hgrid = np.zeros((bh, bw), np.uint8)
    hgrid[::2, :] = 1                                 # horizontal edges at y % 8 == 0 (bs row = y / 4)
    hb[:bw * bh] = hgrid.ravel()
    qp_y = 38
    qp = np.full((W >> 3) * (H >> 3), qp_y, np.int8)
    dbp = np.zeros((((W + 63) // 64) * ((H + 63) // 64), 2), np.int8)
    keep = [_dev(a) for a in (vb, hb, qp, dbp)]
    dm = L.DbkMaps(vertical_bs=keep[0].data_ptr(), horizontal_bs=keep[1].data_ptr(), qp_y_tab=keep[2].data_ptr(), deblock=keep[3].data_ptr(), is_pcm=None,
                   bs_width=bw, min_cb_width=W >> 3, deblock_stride=2, min_pu_width=W >> 2, min_pu_height=H >> 2, width=W, height=H, log2_ctb_size=6,
                   log2_min_cb_size=3, log2_min_pu_size=2, chroma_format_idc=1, cb_qp_offset=0, cr_qp_offset=0)

    # deblocking_filter_CTB (hevc_filter.c:385-470): bS 1, QP 38 on both sides, no offsets -> beta = betatable[38], tc = tctable[38 + 2 (bS - 1)]
    beta, tc = BETA_TABLE[qp_y], TC_TABLE[qp_y]
    for vertical, word, fn in ((1, "vertical", "hevc_v_loop_filter_luma"), (0, "horizontal", "hevc_h_loop_filter_luma")):
        def launch(pic, ex):
            L.dev_deblock_maps(L.planes_of(pic), bd, dm, vertical, st())
        ms, ring = _time(launch, lambda: _smooth_pic(bd, g))
        pic = _smooth_pic(bd, g)
        before = _np(pic[0], bd).copy()
        launch(pic, None)
        torch.cuda.synchronize()
        got = _np(pic[0], bd)
        bad = 0
        for _ in range(N_CHECK):
            if vertical:
                x, y = 8 * int(rng.integers(1, W // 8)), 8 * int(rng.integers(0, H // 8))
                want = before[y:y + 8, x - 8:x + 8].copy()
                orc.deblock_luma(bd, 1, want, 8, 0, beta, (tc, tc), (0, 0), (0, 0))
                bad += not np.array_equal(got[y:y + 8, x - 4:x + 4], want[:, 4:12])
            else:
                x, y = 8 * int(rng.integers(0, W // 8)), 8 * int(rng.integers(1, H // 8))
                want = before[y - 8:y + 8, x:x + 8].copy()
                orc.deblock_luma(bd, 0, want, 0, 8, beta, (tc, tc), (0, 0), (0, 0))
                bad += not np.array_equal(got[y - 4:y + 4, x:x + 8], want[4:12, :])
        out[f"deblock_luma_{word}_from_maps_{bd}bit"] = _row(
            ms, ring, 2 * P * W * H, W * H, bad, N_CHECK,
            f"every {word} 8x8-grid luma edge of eight stacked 4K pictures, parameters derived on the device from the decoder's maps (bS 1, QP 38); "
            f"deblocking_filter_CTB + {fn} (hevc_filter.c:345-581, hevcdsp_template.c:1629-1723)")


def sao_rows(bd, orc, po, g, rng, st, out):
    P = 2 if bd > 8 else 1
    xs, ys = np.meshgrid(np.arange(0, W, 64), np.arange(0, H, 64))
    n = xs.size
    j = np.zeros(n, L.SAO_JOB)
    j["x"], j["y"] = xs.ravel(), ys.ravel()
    j["w"], j["h"] = np.minimum(64, W - j["x"]), np.minimum(64, H - j["y"])
    j["type"], j["klass"] = L.SAO_EDGE, 2
    j["borders"] = (j["x"] == 0) * 1 + (j["y"] == 0) * 2 + (j["x"] + j["w"] == W) * 4 + (j["y"] + j["h"] == H) * 8
    j["offset_val"] = [0, 3, 1, -1, -3]
    d_jobs = _dev(j)

    def launch(pic, ex):
        L.dev_sao_batch_sorted(L.planes_of(pic), L.planes_of(ex), bd, d_jobs.data_ptr(), n, 0, st())
    ms, ring = _time(launch, lambda: _smooth_pic(bd, g), lambda k: _smooth_pic(bd, g), _bytes(_smooth_pic(bd, g)))
    pic, src = _smooth_pic(bd, g), _smooth_pic(bd, g)
    launch(pic, src)
    torch.cuda.synchronize()
    got, s0 = _np(pic[0], bd), _np(src[0], bd)
    bad = 0
    want = np.zeros_like(s0)
    for k in rng.integers(0, n, N_CHECK):
        x, y, w, h, bdr = int(j["x"][k]), int(j["y"][k]), int(j["w"][k]), int(j["h"][k]), int(j["borders"][k])
        orc.sao_edge(bd, 0, want, s0, x, y, w, h, [0, 3, 1, -1, -3], 2, [bdr & 1, (bdr >> 1) & 1, (bdr >> 2) & 1, (bdr >> 3) & 1])
        bad += not np.array_equal(got[y:y + h, x:x + w], want[y:y + h, x:x + w])
    out[f"sao_edge_luma_{bd}bit"] = _row(
        ms, ring, 2 * P * W * H, W * H, bad, N_CHECK,
        f"{n} luma CTBs 64x64, edge offset class 2 (135 degrees), reading a deblocked copy and writing the picture; sao_edge_filter (hevcdsp_template.c:372-567)")


def tu_rows(orc, po, g, rng, st, out):
    """the graded kernel's other configurations (BASELINE config 2): 16x16 at 8 bit (2^22 blocks) and 32x32 at 10 bit (2^20 blocks), as bench.py's
    headline lays them out (16384-sample-wide tiled plane, fresh random prediction plane per launch)"""
    for log2, bd in ((4, 8), (5, 10)):
        n = 1 << log2
        nblk = 1 << (22 if log2 == 4 else 20)
        per_row = 16384 // n
        Hh, Ww = nblk // per_row * n, 16384
        dt = torch.uint8 if bd == 8 else torch.int16
        coeffs = torch.randint(-1024, 1024, (nblk, n, n), dtype=torch.int16, device="cuda", generator=g)
        idx = np.arange(nblk)
        jobs = np.zeros(nblk, L.TU_JOB)
        jobs["x"], jobs["y"], jobs["coeff_off"] = (idx % per_row) * n, (idx // per_row) * n, idx.astype(np.uint32) * n * n
        d_jobs = _dev(jobs)

        def fresh():
            return [torch.randint(0, 1 << bd, (Hh, Ww), dtype=torch.int32, device="cuda", generator=g).to(dt), None, None]

        def launch(pic, ex):
            L.dev_tu_batch(L.planes_of(pic), bd, log2, L.TU_IDCT, d_jobs.data_ptr(), nblk, coeffs.data_ptr(), st())
        global RING_BYTES
        keep, RING_BYTES = RING_BYTES, 3 << 30            # (a plane is 1-2 GiB: three of them, so that no launch meets its own initialisation)
        try:
            ms, ring = _time(launch, fresh)
        finally:
            RING_BYTES = keep
        pic = fresh()
        picks = [(int(k), _np(pic[0][int(jobs["y"][k]):int(jobs["y"][k]) + n, int(jobs["x"][k]):int(jobs["x"][k]) + n], bd).copy()) for k in rng.integers(0, nblk, N_CHECK)]
        launch(pic, None)
        torch.cuda.synchronize()
        bad = 0
        for k, before in picks:
            x, y = int(jobs["x"][k]), int(jobs["y"][k])
            want = orc.tu_batch(bd, po.TU_IDCT, log2, coeffs[k:k + 1].cpu().numpy(), before, np.zeros((1, 2), np.int32))
            bad += not np.array_equal(_np(pic[0][y:y + n, x:x + n], bd), want)
        px = 2 if bd > 8 else 1
        out[f"idct_add_{n}x{n}_{bd}bit"] = _row(
            ms, ring, nblk * n * n * (2 + 2 * px), nblk * n * n, bad, N_CHECK,
            f"{nblk} blocks {n}x{n} inverse DCT + add, {bd}-bit, coefficients U[-1024,1023], 16384-wide tiled plane (BASELINE config 2); idct + transform_add "
            f"(hevcdsp_template.c:45-111,210-301)")
        del coeffs, d_jobs, pic


def run(orc, po, only=None):
    """all rows; `only`: substring filter on the family name (mc / intra / deblock / sao / idct)"""
    L.load_library()
    g = torch.Generator(device="cuda").manual_seed(7)
    rng = np.random.default_rng(7)
    st = lambda: torch.cuda.current_stream().cuda_stream
    out = {}
    fams = [("idct", lambda: tu_rows(orc, po, g, rng, st, out))]
    for bd in (8, 10):
        fams += [("mc", lambda bd=bd: mc_rows(bd, orc, po, g, rng, st, out)), ("intra", lambda bd=bd: intra_rows(bd, orc, po, g, rng, st, out)),
                 ("deblock", lambda bd=bd: deblock_rows(bd, orc, po, g, rng, st, out)), ("sao", lambda bd=bd: sao_rows(bd, orc, po, g, rng, st, out))]
    for name, fn in fams:
        if only is None or only in name:
            try:
                fn()
            except Exception as e:                       # a row that cannot run must not take the bench line with it
                out[f"{name}_error"] = f"{type(e).__name__}: {e}"
            torch.cuda.empty_cache()
    # the rest of the table (chroma / weighted MC, SAO band, chroma deblocking, small transforms, up-sampling, coefficient expansion, boundary
    # strengths): tools/kernel_rows_more.py, same rules
    if only is None or only == "more" or only.startswith("more:"):
        import kernel_rows_more
        kernel_rows_more.run(orc, po, out, None if only in (None, "more") else only[5:])
    return out


if __name__ == "__main__":
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pyoracle as po
    print(json.dumps(run(po.load("oracle"), po, sys.argv[1] if len(sys.argv) > 1 else None), indent=1))
