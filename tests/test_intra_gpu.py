"""-m gpu parity tests of intra prediction (host job resolution + HIP kernel) vs the CPU oracle's intra_pred()."""
import numpy as np
import pytest

from openhevc_amd import lib as L
import gpu_util as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd", [8, 10, 12, 14])
def test_intra_pred_random_calls(oracle, bd):
    rng = np.random.default_rng(900 + bd)
    W, H = 136, 72                                      # not CTB aligned -> picture-edge clipping of the 2N neighbours
    for it in range(250):
        log2 = int(rng.integers(2, 6)); n = 1 << log2
        c_idx = int(rng.integers(0, 3)); cfi = int(rng.choice([1, 1, 1, 3]))
        sh = 1 if (c_idx and cfi == 1) else 0
        nl = n << sh
        x0 = int(rng.integers(0, (W - nl) // nl + 1)) * nl; y0 = int(rng.integers(0, (H - nl) // nl + 1)) * nl
        if it % 5 == 0:
            x0, y0 = (W - nl) // nl * nl, (H - nl) // nl * nl
        mode = int(rng.integers(0, 35)) if it % 3 else int(rng.choice([0, 1, 10, 26, 2, 18, 34]))
        cands = [int(rng.random() < 0.7) for _ in range(5)]
        if x0 == 0: cands[0] = cands[1] = cands[2] = 0
        if y0 == 0: cands[2] = cands[3] = cands[4] = 0
        if x0 + nl >= W: cands[4] = 0
        if y0 + nl >= H: cands[0] = 0
        if it % 7 == 0:                                 # smooth content: exercises the strong 32x32 filter
            planes = [np.ascontiguousarray(np.full((H + 8, W + 8), int(rng.integers(0, 1 << bd)), G.pixdt(bd)) + (np.arange(W + 8) // 16).astype(G.pixdt(bd))) for _ in range(3)]
        else:
            planes = [rng.integers(0, 1 << bd, size=(H + 8, W + 8)).astype(G.pixdt(bd)) for _ in range(3)]
        strong = int(rng.random() < 0.7); dis = int(rng.random() < 0.1); ctb = int(rng.choice([4, 5, 6]))
        want = [p.copy() for p in planes]
        oracle.intra_pred(bd, want, W, H, x0, y0, log2, c_idx, mode, cands, chroma_format_idc=cfi, strong=strong,
                          smoothing_disabled=dis, log2_ctb_size=ctb, log2_min_tb_size=2)
        geom = L.IntraGeom(W, H, cfi, ctb, 2, strong, dis, 0)
        job = L.intra_make_job(geom, x0, y0, log2, c_idx, mode, cands)
        d = [G.to_dev(p) for p in planes]
        d_jobs = G.to_dev(job)
        L.dev_intra_batch(G.planes3(d), bd, d_jobs.data_ptr(), 1, G.stream())
        G.sync()
        for pl in range(3):
            got = G.to_host(d[pl], planes[pl].dtype)
            assert np.array_equal(got, want[pl]), (it, log2, c_idx, mode, cands, x0, y0, cfi, strong, dis, ctb, pl)


def test_intra_batch_of_independent_blocks(oracle):
    """Many non-adjacent blocks in one launch (one wavefront each)."""
    bd, W, H = 8, 640, 320
    rng = np.random.default_rng(4)
    plane = rng.integers(0, 256, size=(H, W)).astype(np.uint8)
    planes = [plane, plane[: H // 2, : W // 2].copy(), plane[: H // 2, : W // 2].copy()]
    want = [p.copy() for p in planes]
    geom = L.IntraGeom(W, H, 1, 6, 2, 1, 0, 0)
    jobs = []
    for cy in range(0, H - 127, 128):
        for cx in range(0, W - 127, 128):
            log2 = int(rng.integers(2, 6)); mode = int(rng.integers(0, 35))
            x0, y0 = cx + 64, cy + 64
            cands = [int(rng.random() < 0.8) for _ in range(5)]
            oracle.intra_pred(bd, want, W, H, x0, y0, log2, 0, mode, cands, chroma_format_idc=1, strong=1, smoothing_disabled=0,
                              log2_ctb_size=6, log2_min_tb_size=2)
            jobs.append(L.intra_make_job(geom, x0, y0, log2, 0, mode, cands)[0])
    batch = np.array(jobs, dtype=L.INTRA_JOB)
    d = [G.to_dev(p) for p in planes]
    d_jobs = G.to_dev(batch)
    L.dev_intra_batch(G.planes3(d), bd, d_jobs.data_ptr(), len(batch), G.stream())
    G.sync()
    assert np.array_equal(G.to_host(d[0], np.uint8), want[0])


@pytest.mark.parametrize("bd", [8, 10, 14])
def test_intra_pred_constrained(oracle, bd):
    """constrained_intra_pred_flag streams: host re-derivation of availability + the kernel's substitution walk."""
    rng = np.random.default_rng(950 + bd)
    W, H = 136, 72
    for it in range(250):
        log2 = int(rng.integers(2, 6)); n = 1 << log2
        c_idx = int(rng.integers(0, 3))
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
        is_intra = (rng.random((ph, pw)) < float(rng.choice([0.0, 0.2, 0.5, 0.8, 1.0]))).astype(np.uint8)
        is_intra[y0 >> lpu:((y0 + nl - 1) >> lpu) + 1, x0 >> lpu:((x0 + nl - 1) >> lpu) + 1] = 1
        planes = [rng.integers(0, 1 << bd, size=(H + 8, W + 8)).astype(G.pixdt(bd)) for _ in range(3)]
        strong = int(rng.random() < 0.7)
        want = [p.copy() for p in planes]
        oracle.intra_pred(bd, want, W, H, x0, y0, log2, c_idx, mode, cands, chroma_format_idc=1, strong=strong, smoothing_disabled=0,
                          log2_ctb_size=6, log2_min_tb_size=2, log2_min_pu_size=lpu, constrained=1, is_intra=is_intra)
        geom = L.IntraGeom(W, H, 1, 6, 2, strong, 0, 1)
        job, cip = L.intra_make_job_cip(geom, lpu, is_intra, x0, y0, log2, c_idx, mode, cands)
        assert job["flags2"][0] & L.INTRA2_CIP
        d = [G.to_dev(p) for p in planes]
        d_jobs = G.to_dev(job); d_cip = G.to_dev(cip)
        L.dev_intra_batch_cip(G.planes3(d), bd, d_jobs.data_ptr(), 1, d_cip.data_ptr(), G.stream())
        G.sync()
        for pl in range(3):
            assert np.array_equal(G.to_host(d[pl], planes[pl].dtype), want[pl]), (it, log2, c_idx, mode, cands, x0, y0, lpu, pl)


@pytest.mark.parametrize("bd", [8, 10, 14])
def test_intra_pred_constrained_structured_maps(oracle, bd):
    """Same check with intra/inter maps made of coding-unit sized patches on a wider picture (what real streams look like:
    long runs of non-intra neighbours next to fully intra ones), all mismatches reported."""
    rng = np.random.default_rng(5 + bd)
    W, H = 448, 232
    bad = []
    for it in range(500):
        log2 = int(rng.integers(2, 6)); n = 1 << log2
        c_idx = int(rng.integers(0, 3)); sh = 1 if c_idx else 0
        nl = n << sh
        x0 = int(rng.integers(0, (W - nl) // nl + 1)) * nl; y0 = int(rng.integers(1, (H - nl) // nl + 1)) * nl
        mode = int(rng.integers(0, 35)) if it % 3 else 1
        cands = [int(rng.random() < 0.85) for _ in range(5)]
        if x0 == 0: cands[0] = cands[1] = cands[2] = 0
        if x0 + nl >= W: cands[4] = 0
        if y0 + nl >= H: cands[0] = 0
        lpu = int(rng.choice([2, 3]))
        pw, ph = (W + (1 << lpu) - 1) >> lpu, (H + (1 << lpu) - 1) >> lpu
        g = int(rng.choice([1, 2, 4, 8, 16]))
        coarse = rng.random((ph // g + 1, pw // g + 1)) < float(rng.choice([0.2, 0.5, 0.8]))
        is_intra = np.ascontiguousarray(np.kron(coarse, np.ones((g, g)))[:ph, :pw].astype(np.uint8))
        is_intra[y0 >> lpu:((y0 + nl - 1) >> lpu) + 1, x0 >> lpu:((x0 + nl - 1) >> lpu) + 1] = 1
        planes = [rng.integers(0, 1 << bd, size=(H + 8, W + 8)).astype(G.pixdt(bd)) for _ in range(3)]
        strong = int(rng.random() < 0.7)
        want = [p.copy() for p in planes]
        oracle.intra_pred(bd, want, W, H, x0, y0, log2, c_idx, mode, cands, chroma_format_idc=1, strong=strong, smoothing_disabled=0,
                          log2_ctb_size=6, log2_min_tb_size=2, log2_min_pu_size=lpu, constrained=1, is_intra=is_intra)
        geom = L.IntraGeom(W, H, 1, 6, 2, strong, 0, 1)
        job, cip = L.intra_make_job_cip(geom, lpu, is_intra, x0, y0, log2, c_idx, mode, cands)
        d = [G.to_dev(p) for p in planes]
        d_jobs = G.to_dev(job); d_cip = G.to_dev(cip)
        L.dev_intra_batch_cip(G.planes3(d), bd, d_jobs.data_ptr(), 1, d_cip.data_ptr(), G.stream())
        G.sync()
        for pl in range(3):
            if not np.array_equal(G.to_host(d[pl], planes[pl].dtype), want[pl]):
                bad.append((it, log2, c_idx, mode, cands, x0, y0, lpu, g, int(job["flags"][0]), int(job["flags2"][0])))
    assert not bad, (len(bad), bad[:8])


# ---------------------------------------------------------------------------------------------------- the packed kernel (N lanes per block)
# ohevc_dev_intra_recon_sorted: jobs sorted by size, 16 / 8 / 4 / 2 blocks per wavefront, substitution rules folded into the neighbour
# addresses, the block's residual added in registers.  Same oracle: intra_pred(), then the residual of the block (tu_batch).
def _random_block(rng, W, H, bd, log2, cfi=1, all_modes=True):
    n = 1 << log2
    c_idx = int(rng.integers(0, 3))
    sh = 1 if (c_idx and cfi == 1) else 0
    nl = n << sh
    x0 = int(rng.integers(0, (W - nl) // nl + 1)) * nl
    y0 = int(rng.integers(0, (H - nl) // nl + 1)) * nl
    mode = int(rng.integers(0, 35)) if all_modes else int(rng.choice([0, 1, 10, 26, 2, 18, 34]))
    cands = [int(rng.random() < 0.7) for _ in range(5)]
    if x0 == 0: cands[0] = cands[1] = cands[2] = 0
    if y0 == 0: cands[2] = cands[3] = cands[4] = 0
    if x0 + nl >= W: cands[4] = 0
    if y0 + nl >= H: cands[0] = 0
    return c_idx, x0, y0, mode, cands


@pytest.mark.parametrize("bd", [8, 10, 12, 14])
def test_intra_pack_random_calls(oracle, bd):
    """one block per launch, prediction only: every size x mode x availability pattern x picture edge, vs intra_pred()"""
    rng = np.random.default_rng(1900 + bd)
    W, H = 136, 72
    for it in range(300):
        log2 = int(rng.integers(2, 6))
        cfi = int(rng.choice([1, 1, 1, 3]))
        c_idx, x0, y0, mode, cands = _random_block(rng, W, H, bd, log2, cfi, all_modes=bool(it % 3))
        if it % 5 == 0:
            nl = (1 << log2) << (1 if (c_idx and cfi == 1) else 0)
            x0, y0 = (W - nl) // nl * nl, (H - nl) // nl * nl
            cands[4] = 0; cands[0] = 0
        if it % 11 == 0:
            cands = [0, 0, 0, 0, 0]
        if it % 7 == 0:                                 # smooth content: exercises the strong 32x32 filter
            planes = [np.ascontiguousarray(np.full((H + 8, W + 8), int(rng.integers(0, 1 << bd)), G.pixdt(bd)) + (np.arange(W + 8) // 16).astype(G.pixdt(bd))) for _ in range(3)]
        else:
            planes = [rng.integers(0, 1 << bd, size=(H + 8, W + 8)).astype(G.pixdt(bd)) for _ in range(3)]
        strong = int(rng.random() < 0.7); dis = int(rng.random() < 0.1); ctb = int(rng.choice([4, 5, 6]))
        want = [p.copy() for p in planes]
        oracle.intra_pred(bd, want, W, H, x0, y0, log2, c_idx, mode, cands, chroma_format_idc=cfi, strong=strong,
                          smoothing_disabled=dis, log2_ctb_size=ctb, log2_min_tb_size=2)
        job = L.intra_make_job(L.IntraGeom(W, H, cfi, ctb, 2, strong, dis, 0), x0, y0, log2, c_idx, mode, cands)
        d = [G.to_dev(p) for p in planes]
        d_jobs = G.to_dev(job)
        counts = [0, 0, 0, 0]; counts[log2 - 2] = 1
        L.dev_intra_recon_sorted(G.planes3(d), bd, d_jobs.data_ptr(), 0, counts, 0, G.stream())
        G.sync()
        for pl in range(3):
            got = G.to_host(d[pl], planes[pl].dtype)
            assert np.array_equal(got, want[pl]), (it, log2, c_idx, mode, cands, x0, y0, cfi, strong, dis, ctb, pl)


@pytest.mark.parametrize("bd", [8, 10, 14])
def test_intra_pack_batch_with_residuals(oracle, bd):
    """many independent blocks of all four sizes in ONE launch, most with a residual of a random kind riding along"""
    import ctypes as C
    rng = np.random.default_rng(2900 + bd)
    W, H = 2048, 1536
    luma = rng.integers(0, 1 << bd, size=(H, W)).astype(G.pixdt(bd))
    planes = [luma, rng.integers(0, 1 << bd, size=(H // 2, W // 2)).astype(G.pixdt(bd)), rng.integers(0, 1 << bd, size=(H // 2, W // 2)).astype(G.pixdt(bd))]
    want = [p.copy() for p in planes]
    geom = L.IntraGeom(W, H, 1, 6, 2, 1, 0, 0)
    kinds = [None, L.TU_IDCT, L.TU_IDCT, L.TU_DC, L.TU_SKIP, L.TU_SKIP_RDPCM_H, L.TU_SKIP_RDPCM_V, L.TU_BYPASS, L.TU_BYPASS_RDPCM_H, L.TU_BYPASS_RDPCM_V]
    recs = {2: [], 3: [], 4: [], 5: []}
    arena = []
    off = 0
    # one block per 128x128 luma cell, away from the cell's border: no block reads what another writes
    for cy in range(0, H, 128):
        for cx in range(0, W, 128):
            log2 = int(rng.integers(2, 6)); n = 1 << log2
            c_idx = int(rng.integers(0, 3))
            sh = 1 if c_idx else 0
            x0, y0 = cx + 64, cy + 64                      # luma position; chroma blocks sit at half of it
            mode = int(rng.integers(0, 35))
            cands = [int(rng.random() < 0.8) for _ in range(5)]
            nl = n << sh                                    # (what lies outside the picture is never a candidate: the decoder derives the flags from
            if y0 + nl >= H:                                #  z-scan availability, hevc.c:1107-1130, and nothing outside the picture is available)
                cands[0] = 0
            if x0 + nl >= W:
                cands[4] = 0
            oracle.intra_pred(bd, want, W, H, x0, y0, log2, c_idx, mode, cands, chroma_format_idc=1, strong=1, smoothing_disabled=0,
                              log2_ctb_size=6, log2_min_tb_size=2)
            job = L.intra_make_job(geom, x0, y0, log2, c_idx, mode, cands)[0]
            kind = kinds[int(rng.integers(0, len(kinds)))]
            if kind == L.TU_IDCT and log2 == 2 and c_idx == 0 and rng.random() < 0.5:
                kind = L.TU_DST4
            res = np.zeros(1, L.TU_JOB)[0]
            if kind is not None:
                cf = rng.integers(-512, 512, size=(1, n, n)).astype(np.int16)
                if kind == L.TU_IDCT and rng.random() < 0.5:
                    cf[:, n // 2:, :] = 0; cf[:, :, n // 2:] = 0
                px, py = x0 >> sh, y0 >> sh
                oracle.tu_batch(bd, kind, log2, cf, want[c_idx], np.array([[px, py]], np.int32))
                res["x"], res["y"], res["plane"], res["reserved0"] = px, py, c_idx, kind + 1
                if kind == L.TU_DC:
                    res["dc"] = cf[0, 0, 0]
                else:
                    res["coeff_off"] = off
                    arena.append(cf.reshape(-1)); off += n * n
            recs[log2].append((job, res))
    jobs = np.array([j for k in (2, 3, 4, 5) for j, _ in recs[k]], dtype=L.INTRA_JOB)
    ress = np.array([r for k in (2, 3, 4, 5) for _, r in recs[k]], dtype=L.TU_JOB)
    counts = [len(recs[k]) for k in (2, 3, 4, 5)]
    d = [G.to_dev(p) for p in planes]
    d_jobs, d_res, d_cf = G.to_dev(jobs), G.to_dev(ress), G.to_dev(np.concatenate(arena))
    L.dev_intra_recon_sorted(G.planes3(d), bd, d_jobs.data_ptr(), d_res.data_ptr(), counts, d_cf.data_ptr(), G.stream())
    G.sync()
    for pl in range(3):
        got = G.to_host(d[pl], planes[pl].dtype)
        bad = np.argwhere(got != want[pl])
        assert bad.size == 0, (pl, counts, bad[:4].tolist())


@pytest.mark.parametrize("bd", [8, 10])
def test_intra_chain_hands_levels_over_including_levels_wider_than_its_workgroup(oracle, bd):
    """ohevc_dev_intra_chain: a run of dependency levels in one launch, ONE workgroup of 8 wavefronts.  Level l + 1 here predicts from what
    level l has just written (blocks of a level sit in one row of 64-sample cells, the next level's right below: its above / above-right
    neighbours are the previous level's rows), levels are 1 .. 40 wavefronts wide - wider than the workgroup, so they take further passes -
    and mix all four block sizes, most blocks with a residual.  Must equal the oracle run level by level (and so one
    ohevc_dev_intra_recon_sorted launch per level)."""
    import ctypes as C
    rng = np.random.default_rng(4100 + bd)
    nlev = 14
    W, H = 16384, 64 + 32 * nlev
    planes = [rng.integers(0, 1 << bd, size=(H, W)).astype(G.pixdt(bd)), rng.integers(0, 1 << bd, size=(H // 2, W // 2)).astype(G.pixdt(bd)),
              rng.integers(0, 1 << bd, size=(H // 2, W // 2)).astype(G.pixdt(bd))]
    want = [p.copy() for p in planes]
    geom = L.IntraGeom(W, H, 1, 6, 2, 1, 0, 0)
    blobs, chain, arena, off = [], [], [], 0
    pos16 = 0                                                # running offset of the staged arrays, in 16-byte units

    def put(a):
        nonlocal pos16
        o = pos16
        b = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        pad = (-b.size) % 256
        blobs.append(np.concatenate([b, np.zeros(pad, np.uint8)]))
        pos16 += (b.size + pad) // 16
        return o

    widths = [1, 3, 8, 9, 17, 40, 2, 16, 24, 5, 33, 1, 12, 7]
    ncells = W // 64
    ycell = np.full((3, ncells), 64)                         # per plane and 64-sample column: the luma row its next block starts at
    for lv in range(nlev):
        recs = {2: [], 3: [], 4: [], 5: []}
        # choose sizes until the level has the wanted number of wavefronts (16 / 8 / 4 / 2 blocks per wavefront)
        cells = rng.permutation(ncells)
        target, k = widths[lv], 0
        def waves():
            return sum((len(recs[q]) + (16 >> (q - 2)) - 1) // (16 >> (q - 2)) for q in (2, 3, 4, 5))
        while waves() < target and k < ncells:
            log2 = int(rng.integers(2, 6)) if target <= 16 else int(rng.choice([3, 4, 5, 5, 5]))      # (a wide level needs blocks that fill wavefronts fast)
            n = 1 << log2
            c_idx = int(rng.integers(0, 3)) if log2 < 5 else 0
            sh = 1 if c_idx else 0
            nl = n << sh
            cell = int(cells[k]); k += 1
            x0 = cell * 64
            y0 = -(-int(ycell[c_idx, cell]) // nl) * nl      # right below the column's previous block (rounded up to the block size)
            if y0 + nl > H:
                continue
            ycell[c_idx, cell] = y0 + nl
            mode = int(rng.integers(0, 35))
            cands = [0, 0, 0, 1, 0]                           # only the row above: written by an EARLIER level, never by this one
            oracle.intra_pred(bd, want, W, H, x0, y0, log2, c_idx, mode, cands, chroma_format_idc=1, strong=1, smoothing_disabled=0, log2_ctb_size=6, log2_min_tb_size=2)
            job = L.intra_make_job(geom, x0, y0, log2, c_idx, mode, cands)[0]
            res = np.zeros(1, L.TU_JOB)[0]
            if rng.random() < 0.8:
                kind = L.TU_DST4 if (log2 == 2 and c_idx == 0 and rng.random() < 0.5) else int(rng.choice([L.TU_IDCT, L.TU_IDCT, L.TU_DC, L.TU_SKIP]))
                cf = rng.integers(-512, 512, size=(1, n, n)).astype(np.int16)
                px, py = x0 >> sh, y0 >> sh
                oracle.tu_batch(bd, kind, log2, cf, want[c_idx], np.array([[px, py]], np.int32))
                res["x"], res["y"], res["plane"], res["reserved0"] = px, py, c_idx, kind + 1
                if kind == L.TU_DC:
                    res["dc"] = cf[0, 0, 0]
                else:
                    res["coeff_off"] = off
                    arena.append(cf.reshape(-1)); off += n * n
            recs[log2].append((job, res))
        jobs = np.array([j for q in (2, 3, 4, 5) for j, _ in recs[q]], dtype=L.INTRA_JOB)
        ress = np.array([r for q in (2, 3, 4, 5) for _, r in recs[q]], dtype=L.TU_JOB)
        counts = [len(recs[q]) for q in (2, 3, 4, 5)]
        fw = [0]
        for q in range(4):
            fw.append(fw[-1] + (counts[q] + (16 >> q) - 1) // (16 >> q))
        assert fw[4] >= min(target, 1)
        chain.append((fw, counts, put(jobs), put(ress)))
    lev = np.zeros(nlev, np.dtype([("first_wave", np.int32, 5), ("njobs", np.int32, 4), ("jobs_off16", np.uint32), ("res_off16", np.uint32), ("reserved", np.int32)]))
    for i, (fw, counts, jo, ro) in enumerate(chain):
        lev[i]["first_wave"], lev[i]["njobs"], lev[i]["jobs_off16"], lev[i]["res_off16"] = fw, counts, jo, ro
    assert lev.itemsize == 48
    assert max(int(l["first_wave"][4]) for l in lev) > 2 * L.load_library().ohevc_intra_chain_workgroup_waves()
    d = [G.to_dev(p) for p in planes]
    d_base, d_lev, d_cf = G.to_dev(np.concatenate(blobs)), G.to_dev(lev), G.to_dev(np.concatenate(arena))
    L.check(L.load_library().ohevc_dev_intra_chain(G.planes3(d), C.c_int(bd), C.c_void_p(d_base.data_ptr()), C.c_void_p(d_lev.data_ptr()), C.c_int(nlev),
                                                   C.c_void_p(d_cf.data_ptr()), C.c_void_p(G.stream())))
    G.sync()
    for pl in range(3):
        got = G.to_host(d[pl], planes[pl].dtype)
        bad = np.argwhere(got != want[pl])
        assert bad.size == 0, (pl, bad[:4].tolist())


@pytest.mark.parametrize("per_level", [1, 64], ids=["1024_levels_of_one_block", "1024_levels_of_four_wavefronts"])
def test_intra_chain_takes_a_whole_launch_of_levels(oracle, per_level):
    """ohevc_intra_chain_max_levels() levels in ONE launch - the size the ctx layer cuts a long chain into (an 8K picture has more levels than
    that).  Round 6 builds a 16-byte descriptor per (level, wavefront slot) in the kernel's prologue, two levels per thread of its 512: with
    exactly 1024 levels nobody owned the END of the last level's slots and the last level ran on garbage - a device fault in the 8K dense
    stream test, nowhere else.  1024 levels of one 4x4 block (the descriptors fit), and of four wavefronts each (4096 slots: more than fit -
    the kernel keeps the record form); every level predicts vertically from the level above."""
    import ctypes as C
    if G.emulating() and per_level > 1:
        pytest.skip("65 536 blocks through the emulator take minutes; the device runs it")
    bd = 8
    lib = L.load_library()
    nlev = int(lib.ohevc_intra_chain_max_levels())
    assert nlev == 1024
    rng = np.random.default_rng(4200 + per_level)
    W, H = 4 * per_level, 64 + 4 * nlev
    planes = [rng.integers(0, 256, size=(H, W)).astype(np.uint8), rng.integers(0, 256, size=(H // 2, max(W // 2, 2))).astype(np.uint8),
              rng.integers(0, 256, size=(H // 2, max(W // 2, 2))).astype(np.uint8)]
    want = [p.copy() for p in planes]
    geom = L.IntraGeom(W, H, 1, 6, 2, 1, 0, 0)
    blobs, pos16 = [], 0

    def put(a):
        nonlocal pos16
        o = pos16
        b = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        pad = (-b.size) % 256
        blobs.append(np.concatenate([b, np.zeros(pad, np.uint8)]))
        pos16 += (b.size + pad) // 16
        return o

    lev = np.zeros(nlev, np.dtype([("first_wave", np.int32, 5), ("njobs", np.int32, 4), ("jobs_off16", np.uint32), ("res_off16", np.uint32), ("reserved", np.int32)]))
    for lv in range(nlev):
        y0 = 64 + 4 * lv
        jobs = []
        for k in range(per_level):
            x0 = 4 * k
            mode = 26 if per_level > 1 else int(rng.choice([26, 0, 1, 30]))      # (vertical only where the oracle loop is long)
            cands = [0, 0, 0, 1, 0]
            oracle.intra_pred(bd, want, W, H, x0, y0, 2, 0, mode, cands, chroma_format_idc=1, strong=1, smoothing_disabled=0, log2_ctb_size=6, log2_min_tb_size=2)
            jobs.append(L.intra_make_job(geom, x0, y0, 2, 0, mode, cands)[0])
        jobs = np.array(jobs, dtype=L.INTRA_JOB)
        nw = (per_level + 15) // 16
        lev[lv]["first_wave"], lev[lv]["njobs"], lev[lv]["jobs_off16"], lev[lv]["res_off16"] = [0, nw, nw, nw, nw], [per_level, 0, 0, 0], put(jobs), 0xffffffff
    d = [G.to_dev(p) for p in planes]
    d_base, d_lev = G.to_dev(np.concatenate(blobs)), G.to_dev(lev)
    L.check(lib.ohevc_dev_intra_chain(G.planes3(d), C.c_int(bd), C.c_void_p(d_base.data_ptr()), C.c_void_p(d_lev.data_ptr()), C.c_int(nlev),
                                      None, C.c_void_p(G.stream())))
    G.sync()
    got = G.to_host(d[0], np.uint8)
    bad = np.argwhere(got != want[0])
    assert bad.size == 0, f"{len(bad)} samples differ, first at {bad[:3].tolist()} (row 64 + 4 x level)"
