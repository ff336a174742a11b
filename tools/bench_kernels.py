#!/usr/bin/env python3
"""Per-kernel throughput of the non-headline families on a 4K (3840x2160) picture's worth of synthetic jobs.
For each kernel: Mpixel/s and algorithmic GB/s (byte models of SURVEY.md 8d) vs the 8 TB/s HBM peak.

--resident (what profiles/r02x_* were made with): every kernel runs over a RING of pictures (destinations, and for motion compensation /
SAO their sources too) that together exceed 2 GiB - eight times the 256 MiB Infinity Cache - written once at set-up; consecutive
launches take consecutive ring entries, so nothing a launch touches is cache-resident from the launch before or from its own
initialisation: the GB/s are HBM numbers.  Pictures are smooth (a ramp plus +-3 of noise) so that the deblocking filter really filters
(on white noise its decision d0 + d3 < beta is almost never true) and SAO classifies real edges.
Without --resident: the round-1 form (a freshly randomised destination right before each launch: Infinity-Cache assisted)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openhevc_amd import lib as L  # noqa: E402

W, H = 3840, 2160
PEAK = 8000.0
# --planes N stacks N 4K pictures vertically into one plane per launch: a single 4K plane is 10-60 us of work for this GPU, i.e.
# mostly launch latency; N = 8 shows what the kernels sustain.  --only SUBSTR keeps the kernels whose name contains SUBSTR.
# --sao-variant V selects the SAO kernel's form (include/ohevc_debug.h: 0 shipped, 1 interior / ring split, 16 interior edge-class blocks
# through the general loop too) and tags the rows; --sao-class K / --sao-width W: the edge class (0 horizontal, 1 vertical, 2 / 3 diagonal)
# and the block width (128: what a workgroup spanning two CTBs would see) of the SAO rows.
# --mc-variant V: ohevc_debug_set_mc_variant (1-6 the kernels; lab build: 102 / 103 mc4q_kernel's traffic-only / arithmetic-only twin, 104 back).
PLANES = 1
ONLY = None
SAO_VARIANT = None
MC_CONFIG = None
SAO_WIDTH = 64             # --sao-width W: blocks of W x 64 samples (what a workgroup spanning two CTBs would see: 128)
SAO_CLASS = 2              # --sao-class K: 0 horizontal, 1 vertical, 2 / 3 the diagonals (edge); the band position (band)
MC_VARIANT = None           # --mc-variant V: 3 = LDS tiles (mc3), 4 = matrix cores (mc4); tags the rows
RESIDENT = "--resident" in sys.argv
RING_BYTES = 2 << 30
for i, a in enumerate(sys.argv):
    if a == "--planes":
        PLANES = int(sys.argv[i + 1])
    if a == "--only":
        ONLY = sys.argv[i + 1]
    if a == "--sao-class":
        SAO_CLASS = int(sys.argv[i + 1])
    if a == "--sao-width":
        SAO_WIDTH = int(sys.argv[i + 1])
    if a == "--sao-variant":
        SAO_VARIANT = int(sys.argv[i + 1])
    if a == "--mc-variant":
        MC_VARIANT = int(sys.argv[i + 1])
    if a == "--mc-config":      # "16,16,0": only this (width, height, bi) of the motion-compensation configurations
        MC_CONFIG = tuple(int(v) for v in sys.argv[i + 1].split(","))
H *= PLANES


def dev(a):
    a = np.ascontiguousarray(a)
    if a.dtype.fields is not None:
        return torch.from_numpy(a.view(np.uint8)).cuda()
    return torch.from_numpy(a).cuda()


def rand_pic(bd, g):
    dt = torch.uint8 if bd == 8 else torch.int16
    if RESIDENT:        # smooth content: a diagonal ramp + small noise (see the module docstring)
        def mk(h, w):
            yy = torch.arange(h, device="cuda", dtype=torch.int32)[:, None]
            xx = torch.arange(w, device="cuda", dtype=torch.int32)[None, :]
            base = ((xx + yy) >> 3) % ((1 << bd) - 16) + 8
            return (base + torch.randint(-3, 4, (h, w), dtype=torch.int32, device="cuda", generator=g)).to(dt)
    else:
        mk = lambda h, w: torch.randint(0, 1 << bd, (h, w), dtype=dt, device="cuda", generator=g)
    return [mk(H, W), mk(H // 2, W // 2), mk(H // 2, W // 2)]


def pic_bytes(pic):
    return sum(t.numel() * t.element_size() for t in pic if t is not None)


def timeit(fn, fresh, reps=12, name="", ring_of=None):
    """ring_of(k): per-entry extra state for entry k of the ring (sources that must rotate with the destination); fn(arg, extra)."""
    if not wanted(name):
        return None
    st = torch.cuda.current_stream()
    ts = []
    if RESIDENT:
        first = fresh()
        n_ring = max(2, -(-RING_BYTES // max(1, pic_bytes(first) + (ring_of.bytes if ring_of else 0))))
        ring = [first] + [fresh() for _ in range(n_ring - 1)]
        extra = [ring_of(k) for k in range(n_ring)] if ring_of else [None] * n_ring
        torch.cuda.synchronize()
        for r in range(reps + 2):
            k = r % n_ring
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st); fn(ring[k], extra[k]); b.record(st)
            torch.cuda.synchronize()
            if r >= 2:
                ts.append(a.elapsed_time(b))
        del ring, extra
        torch.cuda.empty_cache()
        return float(np.median(ts))
    for r in range(reps + 2):
        arg = fresh()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(arg, None); b.record(st)
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(a.elapsed_time(b))
    return float(np.median(ts))


class RingExtra:
    """callable that builds the per-entry sources of a ring (see timeit) and says how many bytes an entry adds"""
    def __init__(self, make, nbytes):
        self.make, self.bytes = make, nbytes

    def __call__(self, k):
        return self.make(k)


def wanted(name):
    return ONLY is None or ONLY in name


def report(name, ms, pixels, alg_bytes, out):
    if PLANES > 1:
        name += f" [x{PLANES} pictures per launch]"
    if RESIDENT:
        name += " [HBM-resident ring]"
    if SAO_VARIANT is not None and "sao" in name:
        name += f" [sao variant {SAO_VARIANT}]"
    if ms is None:
        return
    gbs = alg_bytes / ms / 1e6
    row = {"kernel": name, "ms": round(ms, 4), "Mpixel_per_s": round(pixels / ms / 1e3, 1), "alg_GBps": round(gbs, 1),
           "frac_hbm_peak": round(gbs / PEAK, 4)}
    out.append(row)
    print(json.dumps(row), flush=True)


def main():
    L.load_library()
    if SAO_VARIANT is not None:
        L.load_library().ohevc_debug_set_sao_variant(SAO_VARIANT)
    if MC_VARIANT is not None:
        L.load_library().ohevc_debug_set_mc_variant(MC_VARIANT)
    g = torch.Generator(device="cuda").manual_seed(7)
    rng = np.random.default_rng(7)
    st = lambda: torch.cuda.current_stream().cuda_stream
    out = []
    for bd in (8, 10):
        P = 2 if bd > 8 else 1
        refs = [rand_pic(bd, g) for _ in range(2)]
        table = dev(L.planes_table(refs))
        # ---- MC: tile the luma plane with blocks of one size, random quarter-sample MVs within +-16 samples
        for (bw, bh, bi) in [(8, 8, 0), (16, 16, 0), (16, 16, 1), (32, 32, 1), (64, 64, 1)]:
            if MC_CONFIG is not None and (bw, bh, bi) != MC_CONFIG:
                continue
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
            d_jobs = dev(j)

            def mc_sources(k):          # the two reference pictures of ring entry k and their table
                rr = [rand_pic(bd, g) for _ in range(2)]
                return rr, dev(L.planes_table(rr))
            ms = timeit(lambda pic, ex: L.dev_mc_batch_bounded(L.planes_of(pic), (ex[1] if ex else table).data_ptr(), 2, bd, d_jobs.data_ptr(), n, bw, bh, st()),
                        lambda: rand_pic(bd, g), name="mc", ring_of=RingExtra(mc_sources, 2 * pic_bytes(refs[0])))
            px = n * bw * bh
            alg = n * ((1 + bi) * P * (bw + 7) * (bh + 7) + P * bw * bh)
            tag = "" if MC_VARIANT is None else f" [mc variant {MC_VARIANT}]"
            report(f"mc luma {bw}x{bh} {'bi' if bi else 'uni'} {bd}-bit (random qpel phases){tag}", ms, px, alg, out)
            if bw == 8:             # the same jobs through the small-block entry point (mc3: four jobs per wavefront)
                ms = timeit(lambda pic, ex: L.dev_mc_batch_small(L.planes_of(pic), (ex[1] if ex else table).data_ptr(), 2, bd, d_jobs.data_ptr(), n, st()),
                            lambda: rand_pic(bd, g), name="mc", ring_of=RingExtra(mc_sources, 2 * pic_bytes(refs[0])))
                report(f"mc luma 8x8 uni {bd}-bit, small-block entry point (random qpel phases){tag}", ms, px, alg, out)
        # ---- deblock: every 8x8-grid luma edge of the picture, vertical pass (bS-like params that filter ~always)
        xs, ys = np.meshgrid(np.arange(8, W, 8), np.arange(0, H, 8))
        n = xs.size
        j = np.zeros(n, L.DBK_JOB)
        j["x"], j["y"], j["plane"], j["flags"], j["beta"] = xs.ravel(), ys.ravel(), 0, L.DBK_VERTICAL_EDGE, 40
        j["tc"] = 6
        d_jobs = dev(j)
        ms = timeit(lambda pic, ex: L.dev_deblock_batch(L.planes_of(pic), bd, d_jobs.data_ptr(), n, st()), lambda: rand_pic(bd, g), name="deblock")
        report(f"deblock luma vertical edges, full 4K picture, {bd}-bit", ms, W * H, 2 * P * W * H, out)
        n2 = np.meshgrid(np.arange(0, W, 8), np.arange(8, H, 8))[0].size
        j2 = np.zeros(n2, L.DBK_JOB)
        xs, ys = np.meshgrid(np.arange(0, W, 8), np.arange(8, H, 8))
        j2["x"], j2["y"], j2["plane"], j2["flags"], j2["beta"] = xs.ravel(), ys.ravel(), 0, 0, 40
        j2["tc"] = 6
        d_jobs2 = dev(j2)
        ms = timeit(lambda pic, ex: L.dev_deblock_batch(L.planes_of(pic), bd, d_jobs2.data_ptr(), n2, st()), lambda: rand_pic(bd, g), name="deblock")
        report(f"deblock luma horizontal edges, full 4K picture, {bd}-bit", ms, W * H, 2 * P * W * H, out)
        # ---- deblocking derived on the device from the decoder's maps (ohevc_dev_deblock_maps: what the ctx layer runs): every 8x8-grid luma
        #      edge carries bS 1, chroma planes none (bS 2 only), QP 32 everywhere -> the same luma edges as above, beta 26-ish, tc 1-2
        bw, bh = W >> 2, H >> 2
        vb = np.zeros(bw * (bh + 8), np.uint8); hb = np.zeros((bw + 8) * bh, np.uint8)
        grid = np.zeros((bh, bw), np.uint8); grid[:, ::2] = 1                  # vertical edges at x % 8 == 0 (bs index = x / 4)
        vb[:bw * bh] = grid.ravel()
        grid = np.zeros((bh, bw), np.uint8); grid[::2, :] = 1                  # horizontal edges at y % 8 == 0
        hb[:bw * bh] = grid.ravel()
        qp = np.full((W >> 3) * (H >> 3), 38, np.int8)
        dbp = np.zeros((((W + 63) // 64) * ((H + 63) // 64), 2), np.int8)
        keep = [dev(a) for a in (vb, hb, qp, dbp)]
        dm = L.DbkMaps(vertical_bs=keep[0].data_ptr(), horizontal_bs=keep[1].data_ptr(), qp_y_tab=keep[2].data_ptr(), deblock=keep[3].data_ptr(), is_pcm=None,
                       bs_width=bw, min_cb_width=W >> 3, deblock_stride=2, min_pu_width=W >> 2, min_pu_height=H >> 2, width=W, height=H, log2_ctb_size=6,
                       log2_min_cb_size=3, log2_min_pu_size=2, chroma_format_idc=1, cb_qp_offset=0, cr_qp_offset=0)
        for vert, nm in ((1, "vertical"), (0, "horizontal")):
            ms = timeit(lambda pic, ex: L.dev_deblock_maps(L.planes_of(pic), bd, dm, vert, st()), lambda: rand_pic(bd, g), name="deblock")
            report(f"deblock luma {nm} edges from the decoder's maps, full 4K picture, {bd}-bit", ms, W * H, 2 * P * W * H, out)
        # ---- SAO: one job per 64x64 luma CTB, edge class 2 / band
        src = rand_pic(bd, g)
        for (typ, name) in [(L.SAO_EDGE, "edge (135 deg)"), (L.SAO_BAND, "band")]:
            xs, ys = np.meshgrid(np.arange(0, W, SAO_WIDTH), np.arange(0, H, 64))
            n = xs.size
            j = np.zeros(n, L.SAO_JOB)
            j["x"], j["y"] = xs.ravel(), ys.ravel()
            j["w"], j["h"] = np.minimum(SAO_WIDTH, W - j["x"]), np.minimum(64, H - j["y"])
            j["type"], j["klass"] = typ, SAO_CLASS
            j["borders"] = (j["x"] == 0) * 1 + (j["y"] == 0) * 2 + (j["x"] + j["w"] == W) * 4 + (j["y"] + j["h"] == H) * 8
            j["offset_val"] = [0, 3, 1, -1, -3]
            d_jobs = dev(j)
            # every block here is 64 x 64 and aligned: what the ctx layer does with such a picture (jobs sorted, one kernel)
            run_sao = (lambda pic, ex: L.dev_sao_batch_sorted(L.planes_of(pic), L.planes_of(ex if ex else src), bd, d_jobs.data_ptr(), n, 0, st())) if not ((SAO_VARIANT or 0) & 3) else \
                      (lambda pic, ex: L.dev_sao_batch(L.planes_of(pic), L.planes_of(ex if ex else src), bd, d_jobs.data_ptr(), n, st()))
            ms = timeit(run_sao, lambda: rand_pic(bd, g),
                        name="sao", ring_of=RingExtra(lambda k: rand_pic(bd, g), pic_bytes(src)))
            report(f"sao {name} class {SAO_CLASS} width {SAO_WIDTH} luma, full 4K picture, {bd}-bit", ms, W * H, 2 * P * W * H, out)
        # ---- intra: independent blocks on a sparse grid (every other block position), all 35 modes
        for log2 in (2, 3, 4, 5):
            nn = 1 << log2
            xs, ys = np.meshgrid(np.arange(nn, W - 2 * nn, 2 * nn), np.arange(nn, H - 2 * nn, 2 * nn))
            n = xs.size
            j = np.zeros(n, L.INTRA_JOB)
            j["x"], j["y"], j["log2_size"], j["mode"] = xs.ravel(), ys.ravel(), log2, rng.integers(0, 35, n)
            j["flags"] = 31 | L.INTRA_STRONG | L.INTRA_LUMA_EDGE
            j["bottom_left_size"] = nn; j["top_right_size"] = nn
            d_jobs = dev(j)
            ms = timeit(lambda pic, ex: L.dev_intra_batch(L.planes_of(pic), bd, d_jobs.data_ptr(), n, st()), lambda: rand_pic(bd, g), name="intra")
            report(f"intra {nn}x{nn} independent blocks, one wavefront per block (round-1 kernel), {bd}-bit", ms, n * nn * nn, n * (P * (4 * nn + 1) + P * nn * nn), out)
            # the packed kernel (N lanes per block; what the ctx layer launches per dependency level): prediction only, then with the block's
            # own inverse-DCT residual added in registers (+ 2 N^2 coefficient bytes per block)
            counts = [0, 0, 0, 0]; counts[log2 - 2] = n
            ms = timeit(lambda pic, ex: L.dev_intra_recon_sorted(L.planes_of(pic), bd, d_jobs.data_ptr(), 0, counts, 0, st()), lambda: rand_pic(bd, g), name="intra")
            report(f"intra {nn}x{nn} independent blocks, packed kernel, {bd}-bit", ms, n * nn * nn, n * (P * (4 * nn + 1) + P * nn * nn), out)
            r = np.zeros(n, L.TU_JOB)
            r["x"], r["y"], r["reserved0"], r["coeff_off"] = j["x"], j["y"], L.TU_IDCT + 1, np.arange(n, dtype=np.uint32) * nn * nn
            d_res = dev(r)
            cf = torch.randint(-256, 256, (n * nn * nn,), dtype=torch.int16, device="cuda", generator=g)
            ms = timeit(lambda pic, ex: L.dev_intra_recon_sorted(L.planes_of(pic), bd, d_jobs.data_ptr(), d_res.data_ptr(), counts, cf.data_ptr(), st()),
                        lambda: rand_pic(bd, g), name="intra")
            report(f"intra {nn}x{nn} + inverse DCT residual, packed kernel, {bd}-bit", ms, n * nn * nn, n * (P * (4 * nn + 1) + P * nn * nn + 2 * nn * nn), out)
            del d_res, cf
        # ---- small / special residual kinds
        for (log2, kind, name) in [(2, L.TU_IDCT, "idct4x4"), (2, L.TU_DST4, "dst4x4"), (3, L.TU_IDCT, "idct8x8"), (4, L.TU_DC, "dc16x16"), (3, L.TU_SKIP, "skip8x8")]:
            nn = 1 << log2
            xs, ys = np.meshgrid(np.arange(0, W, nn), np.arange(0, H, nn))
            n = xs.size
            j = np.zeros(n, L.TU_JOB)
            j["x"], j["y"], j["coeff_off"], j["dc"] = xs.ravel(), ys.ravel(), np.arange(n, dtype=np.uint32) * nn * nn, rng.integers(-500, 500, n)
            d_jobs = dev(j)
            coeffs = torch.randint(-1024, 1024, (n * nn * nn,), dtype=torch.int16, device="cuda", generator=g)

            def pic_aligned():
                dt = torch.uint8 if bd == 8 else torch.int16
                return [torch.randint(0, 1 << bd, (H, W), dtype=dt, device="cuda", generator=g), None, None]
            ms = timeit(lambda pic, ex: L.dev_tu_batch(L.planes_of(pic), bd, log2, kind, d_jobs.data_ptr(), n, coeffs.data_ptr(), st()), pic_aligned, name="tu")
            alg = n * ((0 if kind == L.TU_DC else 2 * nn * nn) + 2 * P * nn * nn)
            report(f"tu {name} full 4K luma plane, {bd}-bit", ms, W * H, alg, out)
        # ---- SHVC inter-layer up-sampling: a 1080p base-layer luma plane into the 4K picture (x2, general filter rules)
        bw, bh = W // 2, H // 2
        up = [2048, 2048, 32768, 32768, 2048, 10240, 32768, 32768, 0]
        prm = L.upsample_params(W, H, bw, bh, (0, 0, 0, 0), up, 0)
        cols, col_of, rows, sc, sr = L.upsample_maps(prm, 0)
        d_cols, d_colof, d_rows = dev(cols), dev(col_of), dev(rows)
        dt = torch.uint8 if bd == 8 else torch.int16
        base = torch.randint(0, 1 << bd, (bh, bw), dtype=dt, device="cuda", generator=g)
        ms = timeit(lambda pic, ex: L.dev_upsample_plane(pic[0], ex if ex is not None else base, bd, 0, d_cols.data_ptr(), d_colof.data_ptr(), d_rows.data_ptr(), sc, sr, st()),
                    lambda: rand_pic(bd, g), name="shvc",
                    ring_of=RingExtra(lambda k: torch.randint(0, 1 << bd, (bh, bw), dtype=dt, device="cuda", generator=g), bw * bh * P))
        report(f"shvc upsample x2 luma {bw}x{bh} -> {W}x{H}, {bd}-bit", ms, W * H, P * W * H + P * bw * bh, out)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"bench_kernels_x{PLANES}{'_resident' if RESIDENT else ''}.json"), "w"), indent=1)


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    main()
