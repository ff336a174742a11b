#!/usr/bin/env python3
"""Per-kernel throughput of the non-headline families on a 4K (3840x2160) picture's worth of synthetic jobs.
For each kernel: Mpixel/s and algorithmic GB/s (byte models of SURVEY.md 8d) vs the 8 TB/s HBM peak.  Every timed launch
works on a freshly randomised destination (see bench.py for why)."""
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
# --sao-variant V selects the SAO kernel's form (include/ohevc_debug.h: 0 shipped, 1 interior / ring split) and tags the rows.
PLANES = 1
ONLY = None
SAO_VARIANT = None
for i, a in enumerate(sys.argv):
    if a == "--planes":
        PLANES = int(sys.argv[i + 1])
    if a == "--only":
        ONLY = sys.argv[i + 1]
    if a == "--sao-variant":
        SAO_VARIANT = int(sys.argv[i + 1])
H *= PLANES


def dev(a):
    a = np.ascontiguousarray(a)
    if a.dtype.fields is not None:
        return torch.from_numpy(a.view(np.uint8)).cuda()
    return torch.from_numpy(a).cuda()


def rand_pic(bd, g):
    dt = torch.uint8 if bd == 8 else torch.int16
    mk = lambda h, w: torch.randint(0, 1 << bd, (h, w), dtype=dt, device="cuda", generator=g)
    return [mk(H, W), mk(H // 2, W // 2), mk(H // 2, W // 2)]


def timeit(fn, fresh, reps=12, name=""):
    if not wanted(name):
        return None
    st = torch.cuda.current_stream()
    ts = []
    for r in range(reps + 2):
        arg = fresh()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(arg); b.record(st)
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def wanted(name):
    return ONLY is None or ONLY in name


def report(name, ms, pixels, alg_bytes, out):
    if PLANES > 1:
        name += f" [x{PLANES} pictures per launch]"
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
            ms = timeit(lambda pic: L.dev_mc_batch(L.planes_of(pic), table.data_ptr(), 2, bd, d_jobs.data_ptr(), n, st()), lambda: rand_pic(bd, g), name="mc")
            px = n * bw * bh
            alg = n * ((1 + bi) * P * (bw + 7) * (bh + 7) + P * bw * bh)
            report(f"mc luma {bw}x{bh} {'bi' if bi else 'uni'} {bd}-bit (random qpel phases)", ms, px, alg, out)
        # ---- deblock: every 8x8-grid luma edge of the picture, vertical pass (bS-like params that filter ~always)
        xs, ys = np.meshgrid(np.arange(8, W, 8), np.arange(0, H, 8))
        n = xs.size
        j = np.zeros(n, L.DBK_JOB)
        j["x"], j["y"], j["plane"], j["flags"], j["beta"] = xs.ravel(), ys.ravel(), 0, L.DBK_VERTICAL_EDGE, 40
        j["tc"] = 6
        d_jobs = dev(j)
        ms = timeit(lambda pic: L.dev_deblock_batch(L.planes_of(pic), bd, d_jobs.data_ptr(), n, st()), lambda: rand_pic(bd, g), name="deblock")
        report(f"deblock luma vertical edges, full 4K picture, {bd}-bit", ms, W * H, 2 * P * W * H, out)
        n2 = np.meshgrid(np.arange(0, W, 8), np.arange(8, H, 8))[0].size
        j2 = np.zeros(n2, L.DBK_JOB)
        xs, ys = np.meshgrid(np.arange(0, W, 8), np.arange(8, H, 8))
        j2["x"], j2["y"], j2["plane"], j2["flags"], j2["beta"] = xs.ravel(), ys.ravel(), 0, 0, 40
        j2["tc"] = 6
        d_jobs2 = dev(j2)
        ms = timeit(lambda pic: L.dev_deblock_batch(L.planes_of(pic), bd, d_jobs2.data_ptr(), n2, st()), lambda: rand_pic(bd, g), name="deblock")
        report(f"deblock luma horizontal edges, full 4K picture, {bd}-bit", ms, W * H, 2 * P * W * H, out)
        # ---- SAO: one job per 64x64 luma CTB, edge class 2 / band
        src = rand_pic(bd, g)
        for (typ, name) in [(L.SAO_EDGE, "edge (135 deg)"), (L.SAO_BAND, "band")]:
            xs, ys = np.meshgrid(np.arange(0, W, 64), np.arange(0, H, 64))
            n = xs.size
            j = np.zeros(n, L.SAO_JOB)
            j["x"], j["y"] = xs.ravel(), ys.ravel()
            j["w"], j["h"] = np.minimum(64, W - j["x"]), np.minimum(64, H - j["y"])
            j["type"], j["klass"] = typ, 2
            j["borders"] = (j["x"] == 0) * 1 + (j["y"] == 0) * 2 + (j["x"] + j["w"] == W) * 4 + (j["y"] + j["h"] == H) * 8
            j["offset_val"] = [0, 3, 1, -1, -3]
            d_jobs = dev(j)
            ms = timeit(lambda pic: L.dev_sao_batch(L.planes_of(pic), L.planes_of(src), bd, d_jobs.data_ptr(), n, st()), lambda: rand_pic(bd, g), name="sao")
            report(f"sao {name} luma, full 4K picture, {bd}-bit", ms, W * H, 2 * P * W * H, out)
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
            ms = timeit(lambda pic: L.dev_intra_batch(L.planes_of(pic), bd, d_jobs.data_ptr(), n, st()), lambda: rand_pic(bd, g), name="intra")
            report(f"intra {nn}x{nn} independent blocks, {bd}-bit", ms, n * nn * nn, n * (P * (4 * nn + 1) + P * nn * nn), out)
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
            ms = timeit(lambda pic: L.dev_tu_batch(L.planes_of(pic), bd, log2, kind, d_jobs.data_ptr(), n, coeffs.data_ptr(), st()), pic_aligned, name="tu")
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
        ms = timeit(lambda pic: L.dev_upsample_plane(pic[0], base, bd, 0, d_cols.data_ptr(), d_colof.data_ptr(), d_rows.data_ptr(), sc, sr, st()),
                    lambda: rand_pic(bd, g), name="shvc")
        report(f"shvc upsample x2 luma {bw}x{bh} -> {W}x{H}, {bd}-bit", ms, W * H, P * W * H + P * bw * bh, out)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"bench_kernels_x{PLANES}.json"), "w"), indent=1)


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    main()
