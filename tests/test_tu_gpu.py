"""-m gpu parity tests of the residual (TU) family: HIP path through the C ABI vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

KINDS_ANY = [po.TU_DC, po.TU_SKIP, po.TU_SKIP_RDPCM_H, po.TU_SKIP_RDPCM_V, po.TU_BYPASS, po.TU_BYPASS_RDPCM_H, po.TU_BYPASS_RDPCM_V]


def grid_xy(nblk, n, per_row, x0=0, y0=0):
    return np.array([[x0 + (i % per_row) * n, y0 + (i // per_row) * n] for i in range(nblk)], np.int32)


def check_batch(oracle, bd, log2, kind, nblk, amp, seed, per_row=7, pixel_range=None):
    import gpu_util as G
    rng = np.random.default_rng(seed)
    n = 1 << log2
    rows = (nblk + per_row - 1) // per_row
    W = ((per_row * n + 16 + 15) // 16) * 16
    plane = rng.integers(0, pixel_range or (1 << bd), size=(max(rows, 1) * n + 4, W)).astype(G.pixdt(bd))
    xy = grid_xy(nblk, n, per_row, x0=0, y0=0)
    coeffs = rng.integers(-amp, amp, size=(nblk, n, n)).astype(np.int16)
    dcs = coeffs[:, 0, 0].copy() if kind == po.TU_DC else None
    jobs = G.make_tu_jobs(xy, n, dcs=dcs)
    want = oracle.tu_batch(bd, kind, log2, coeffs, plane.copy(), xy) if nblk else plane.copy()
    got = G.run_tu(bd, log2, kind, [plane], jobs, coeffs)[0]
    bad = np.argwhere(got != want)
    assert bad.size == 0, f"bd={bd} log2={log2} kind={kind} nblk={nblk}: {len(bad)} mismatching samples, first at {bad[:4].tolist()}"


@pytest.mark.parametrize("bd", [8, 10, 12, 14])
@pytest.mark.parametrize("log2", [2, 3, 4, 5])
def test_idct_add_bit_exact(oracle, bd, log2):
    for nblk, amp in [(1, 1024), (3, 1 << 15), (64, 1024), (257, 4096), (1000, 200)]:
        check_batch(oracle, bd, log2, po.TU_IDCT, nblk, amp, seed=bd * 100 + log2 * 10 + nblk)


PRODUCT_FORMS = [16 + 128, 16 + 128 + 1024, 2048 + 16 + 128 + 1024]           # dot2 strips, + nt loads, matrix-core tiles (shipped for 32x32)
LAB_FORMS = [512 + 128, 512 + 128 + 1024, 2048 + 4096 + 144, 2048 + 16384 + 144]   # lab build only (ohevc_debug.h)


@pytest.mark.parametrize("variant", PRODUCT_FORMS + LAB_FORMS)
@pytest.mark.parametrize("bd", [8, 10, 14])
def test_idct_add_epilogue_forms_bit_exact(oracle, bd, variant):
    """The A/B forms of the 16x16 / 32x32 kernel (ohevc_debug.h): wave-private 64-sample strips (16), workgroup-wide 256-sample strips
    (512), non-temporal coefficient loads (1024) -- same pictures, including ragged tails (block counts that leave waves idle) and
    blocks that are not horizontal neighbours."""
    import gpu_util as _G
    if _G.emulating() and bd != 8 and variant in LAB_FORMS:
        pytest.skip("the lab forms run through the emulator at 8 bit only (CPU-suite time); the device runs all of them")
    from openhevc_amd import lib as L
    lib = L.load_library()
    if variant in LAB_FORMS and not lib.ohevc_debug_has_lab():
        pytest.skip("measurement kernel: lab build only")
    old = lib.ohevc_debug_set_tu_variant(variant)
    old_wgs = lib.ohevc_debug_set_tu_pipe_workgroups(3)      # the lab loop forms: workgroups walk several tiles each
    try:
        for log2 in (4, 5):
            for nblk, amp, per_row in [(1, 1024, 7), (3, 1 << 15, 7), (9, 4096, 3), (64, 1024, 8), (257, 4096, 7), (1000, 200, 16), (48, 1 << 15, 8)]:
                check_batch(oracle, bd, log2, po.TU_IDCT, nblk, amp, seed=variant + bd * 100 + log2 * 10 + nblk, per_row=per_row)
    finally:
        lib.ohevc_debug_set_tu_variant(old)
        lib.ohevc_debug_set_tu_pipe_workgroups(old_wgs)


@pytest.fixture(params=[0, 512, 1024, 1536])
def mfma_variant(request):
    """32x32 blocks through the matrix-core kernel (ohevc_debug.h: bit 8 of the TU variant; bits 9 / 10 = its A/B forms)."""
    from openhevc_amd import lib as L
    lib = L.load_library()
    if not lib.ohevc_debug_has_lab():
        pytest.skip("first matrix-core form: lab build only")
    old = lib.ohevc_debug_set_tu_variant(16 + 128 + 256 + request.param)
    yield
    lib.ohevc_debug_set_tu_variant(old)


@pytest.mark.parametrize("bd", [8, 10, 12, 14])
def test_idct32_matrix_core_form_bit_exact(oracle, mfma_variant, bd):
    import gpu_util as G
    from openhevc_amd import lib as L
    assert L.load_library().ohevc_tu_kernel_name(bd, 5, po.TU_IDCT) == b"tu_idct32_mfma_kernel"
    # (the grid-stride loop of the kernel needs more block pairs than its 2048 x 4 waves to turn over: 20001 blocks, once)
    for nblk, amp in [(1, 1024), (2, 1 << 15), (3, 1 << 15), (64, 1024), (257, 4096)] + ([(20001, 1 << 15)] if bd == 8 and not G.emulating() else []):
        check_batch(oracle, bd, 5, po.TU_IDCT, nblk, amp, seed=bd * 1000 + nblk, per_row=7 if nblk < 5000 else 64)


def test_idct32_matrix_core_form_extremes(oracle, mfma_variant):
    import gpu_util as G
    n = 32
    pats = [np.full((n, n), 32767), np.full((n, n), -32768), np.where(np.indices((n, n)).sum(0) % 2, 32767, -32768), np.zeros((n, n)),
            np.where(np.indices((n, n))[0] % 2, -32768, 32767), np.zeros((n, n)), np.zeros((n, n))]
    pats[3][0, 0] = 32767; pats[5][31, 31] = -32768; pats[6][17, 5] = 255
    coeffs = np.stack(pats).astype(np.int16)
    for bd in (8, 10):
        plane = np.random.default_rng(3).integers(0, 1 << bd, size=(n, len(pats) * n)).astype(G.pixdt(bd))
        xy = grid_xy(len(pats), n, len(pats))
        want = oracle.tu_batch(bd, po.TU_IDCT, 5, coeffs, plane.copy(), xy)
        got = G.run_tu(bd, 5, po.TU_IDCT, [plane], G.make_tu_jobs(xy, n), coeffs)[0]
        bad = np.argwhere(got != want)
        assert bad.size == 0, (bd, len(bad), bad[:6].tolist())


@pytest.mark.parametrize("bd", [8, 10, 12, 14])
def test_idct_extreme_coefficients(oracle, bd):
    """All-max / all-min / alternating / single-coefficient blocks drive both clip_int16 stages and the pixel clip -- twelve blocks, so that
    the 32x32 case goes through the shipped matrix-core tile kernel (8 blocks per workgroup: int16 inputs as two int8 planes, where
    32767 and -32768 are the corners of the split) and through the dot2 form that takes what is left of a batch."""
    import gpu_util as G
    for log2 in (2, 3, 4, 5):
        n = 1 << log2
        ii = np.indices((n, n))
        pats = [np.full((n, n), 32767), np.full((n, n), -32768), np.where(ii.sum(0) % 2, 32767, -32768), np.zeros((n, n)),
                np.where(ii[0] % 2, -32768, 32767), np.where(ii[1] % 2, 32640, -129), np.full((n, n), 32639), np.full((n, n), -32641),
                np.zeros((n, n)), np.zeros((n, n)), np.full((n, n), 127), np.full((n, n), -128)]
        pats[3][0, 0] = 32767
        pats[8][n - 1, n - 1] = -32768
        pats[9][n // 2, 1] = 255
        coeffs = np.stack(pats).astype(np.int16)
        plane = np.random.default_rng(3).integers(0, 1 << bd, size=(n, len(pats) * n)).astype(G.pixdt(bd))
        xy = grid_xy(len(pats), n, len(pats))
        want = oracle.tu_batch(bd, po.TU_IDCT, log2, coeffs, plane.copy(), xy)
        got = G.run_tu(bd, log2, po.TU_IDCT, [plane], G.make_tu_jobs(xy, n), coeffs)[0]
        assert np.array_equal(got, want), (bd, log2)


@pytest.mark.parametrize("bd", [8, 10, 12, 14])
def test_dst_and_other_kinds(oracle, bd):
    check_batch(oracle, bd, 2, po.TU_DST4, 333, 1 << 15, seed=bd)
    check_batch(oracle, bd, 2, po.TU_DST4, 5, 500, seed=bd + 1)
    for log2 in (2, 3, 4, 5):
        for kind in KINDS_ANY:
            check_batch(oracle, bd, log2, kind, 37, 1 << 15, seed=bd * 7 + log2 + kind)
            check_batch(oracle, bd, log2, kind, 130, 300, seed=bd * 7 + log2 + kind + 1)


def test_prediction_samples_above_the_legal_range(oracle):
    """16-bit pixel storage can hold values no legal picture has; the reference's constrained-intra substitution produces
    them (0x8080 fill, hevcpred_template.c:159-161) and transform_add reads dst as uint16 (hevcdsp_template.c:45-111):
    clip(pred + res) must treat the prediction as unsigned up to 65535."""
    for bd in (10, 12):
        for log2 in (2, 3, 4, 5):
            for kind in [po.TU_IDCT, po.TU_DC, po.TU_SKIP, po.TU_BYPASS_RDPCM_H] + ([po.TU_DST4] if log2 == 2 else []):
                check_batch(oracle, bd, log2, kind, 61, 1 << 15, seed=bd + log2 + kind, pixel_range=1 << 16)
                check_batch(oracle, bd, log2, kind, 61, 600, seed=bd + log2 + kind + 50, pixel_range=1 << 16)


@pytest.mark.parametrize("bd", [8, 10, 12, 14])
def test_cross_component_prediction(oracle, bd):
    """OHEVC_TU_CROSS: chroma residual = own residual + (res_scale_val * luma residual) >> 3 with every pairing of residual
    kinds, including chroma blocks without coded coefficients (hevc.c:1291-1365, hevc_cabac.c:1942-1949)."""
    import gpu_util as G
    from openhevc_amd import lib as L
    rng = np.random.default_rng(40 + bd)
    kinds_any = [po.TU_IDCT, po.TU_DC, po.TU_SKIP, po.TU_SKIP_RDPCM_H, po.TU_SKIP_RDPCM_V, po.TU_BYPASS, po.TU_BYPASS_RDPCM_H,
                 po.TU_BYPASS_RDPCM_V]
    for log2 in (2, 3, 4, 5):
        n = 1 << log2
        nblk, per_row = 45, 9
        planes = [rng.integers(0, 1 << bd, size=(5 * n, ((per_row * n + 31) // 16) * 16)).astype(G.pixdt(bd)) for _ in range(3)]
        want = [p.copy() for p in planes]
        jobs = np.zeros(nblk, L.TU_JOB)
        arena = []
        for i in range(nblk):
            x, y, pl = (i % per_row) * n, (i // per_row) * n, 1 + i % 2
            ky = int(rng.choice(kinds_any + ([po.TU_DST4] if log2 == 2 else [])))
            kc = None if i % 5 == 0 else int(rng.choice(kinds_any))
            amp = int(rng.choice([30, 600, 1 << 15]))
            cy = rng.integers(-amp, amp, size=(n, n)).astype(np.int16)
            cc = rng.integers(-amp, amp, size=(n, n)).astype(np.int16)
            scale = int(rng.choice([1, 2, 4, 8])) * int(rng.choice([-1, 1]))
            oracle.tu_cross(bd, log2, kc, cc, ky, cy, scale, want[pl], x, y)
            jobs[i]["x"], jobs[i]["y"], jobs[i]["plane"] = x, y, pl
            jobs[i]["reserved0"] = (15 if kc is None else kc) | (ky << 4)
            jobs[i]["dc"] = scale
            jobs[i]["reserved1"] = len(arena) * n * n
            arena.append(cy)
            if kc is not None:
                jobs[i]["coeff_off"] = len(arena) * n * n
                arena.append(cc)
        got = G.run_tu(bd, log2, L.TU_CROSS, planes, jobs, np.stack(arena))
        for pl in range(3):
            bad = np.argwhere(got[pl] != want[pl])
            assert bad.size == 0, f"bd={bd} log2={log2} plane={pl}: {len(bad)} mismatches, first {bad[:3].tolist()}"


def test_empty_batch_and_three_planes(oracle):
    import gpu_util as G
    check_batch(oracle, 8, 5, po.TU_IDCT, 0, 100, seed=1)
    rng = np.random.default_rng(9)
    n, bd = 16, 10
    planes = [rng.integers(0, 1 << bd, size=(4 * n, 8 * n)).astype(np.uint16) for _ in range(3)]
    xy = grid_xy(30, n, 8)
    pl = rng.integers(0, 3, size=30).astype(np.uint8)
    coeffs = rng.integers(-2048, 2048, size=(30, n, n)).astype(np.int16)
    want = [p.copy() for p in planes]
    for i in range(30):
        oracle.tu_batch(bd, po.TU_IDCT, 4, coeffs[i:i + 1], want[pl[i]], xy[i:i + 1])
    got = G.run_tu(bd, 4, po.TU_IDCT, planes, G.make_tu_jobs(xy, n, planes=pl), coeffs)
    for i in range(3):
        assert np.array_equal(got[i], want[i]), i


def test_sparse_coefficients_match_limited_reference_transform(oracle):
    """Decoder-legal input (zeros beyond col_limit): the GPU's full transform equals the reference's limited one."""
    import gpu_util as G
    rng = np.random.default_rng(77)
    for log2 in (3, 4, 5):
        n = 1 << log2
        for col_limit in (4, 8, 12, 24):
            if col_limit > n:
                continue
            yy, xx = np.mgrid[0:n, 0:n]
            coeffs = np.where(xx + yy <= col_limit - 4, rng.integers(-3000, 3000, size=(20, n, n)), 0).astype(np.int16)
            plane = rng.integers(0, 256, size=(4 * n, 5 * n + (16 - (5 * n) % 16) % 16)).astype(np.uint8)
            xy = grid_xy(20, n, 5)
            want = oracle.tu_batch(8, po.TU_IDCT, log2, coeffs, plane.copy(), xy, col_limit=col_limit)
            got = G.run_tu(8, log2, po.TU_IDCT, [plane], G.make_tu_jobs(xy, n), coeffs)[0]
            assert np.array_equal(got, want), (log2, col_limit)


def test_full_size_batch_sampled_against_oracle(oracle):
    """BASELINE config 2 scale (2^20 blocks of 32x32 on a 16384-wide tiled plane): blocks are independent, so a
    random sample of blocks is compared with the oracle and a zero-coefficient region must stay untouched."""
    import torch
    import gpu_util as G
    from openhevc_amd import lib as L
    n, nblk, per_row = 32, 1 << 20, 512
    g = torch.Generator(device="cuda").manual_seed(1234)
    plane = torch.randint(0, 256, (nblk // per_row * n, per_row * n), dtype=torch.uint8, device="cuda", generator=g)
    coeffs = torch.randint(-1024, 1024, (nblk, n, n), dtype=torch.int16, device="cuda", generator=g)
    coeffs[1000:2000] = 0
    before = plane.clone()
    idx = np.arange(nblk)
    jobs = np.zeros(nblk, L.TU_JOB)
    jobs["x"], jobs["y"], jobs["coeff_off"] = (idx % per_row) * n, (idx // per_row) * n, idx.astype(np.uint32) * n * n
    d_jobs = G.to_dev(jobs)
    L.dev_tu_batch(L.planes_of([plane, None, None]), 8, 5, L.TU_IDCT, d_jobs.data_ptr(), nblk, coeffs.data_ptr(),
                   torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    sample = np.concatenate([np.random.default_rng(5).choice(nblk, 2048, replace=False), [0, nblk - 1, 1500]])
    for b in sample:
        x, y = int(jobs["x"][b]), int(jobs["y"][b])
        ref_blk = before[y:y + n, x:x + n].cpu().numpy()
        oracle.tu_batch(8, po.TU_IDCT, 5, coeffs[b].cpu().numpy(), ref_blk, np.array([[0, 0]], np.int32))
        assert np.array_equal(plane[y:y + n, x:x + n].cpu().numpy(), ref_blk), b
    ys = slice((1000 // per_row + 1) * n, (2000 // per_row) * n)
    assert torch.equal(plane[ys], before[ys])


def test_multi_segment_launch_equals_per_bin_launches(oracle):
    """ohevc_dev_tu_multi: a mix of sizes and kinds in one launch == the oracle applied bin by bin."""
    import ctypes as C
    import gpu_util as G
    from openhevc_amd import lib as L
    rng = np.random.default_rng(2024)
    bd = 10
    plane = rng.integers(0, 1 << bd, size=(256, 512)).astype(np.uint16)
    want = plane.copy()
    bins = [(5, po.TU_IDCT, 7), (4, po.TU_IDCT, 9), (3, po.TU_IDCT, 33), (2, po.TU_IDCT, 50), (2, po.TU_DST4, 21),
            (4, po.TU_DC, 5), (3, po.TU_SKIP, 11), (2, po.TU_BYPASS_RDPCM_V, 3), (5, po.TU_BYPASS, 2)]
    # non-overlapping placement: one 32-row band per bin
    jobs_all, coeffs_all, segs = [], [], []
    coff = 0
    for band, (log2, kind, n) in enumerate(bins[:8]):
        pass
    y0 = 0
    for (log2, kind, n) in bins:
        nn = 1 << log2
        per_row = 512 // nn
        xy = np.array([[(i % per_row) * nn, y0 + (i // per_row) * nn] for i in range(n)], np.int32)
        c = rng.integers(-1500, 1500, size=(n, nn, nn)).astype(np.int16)
        oracle.tu_batch(bd, kind, log2, c, want, xy)
        j = np.zeros(n + (-n) % 16, L.TU_JOB)                      # pad every bin to a 16-job (256-byte) boundary like the ctx layer does
        j["x"][:n], j["y"][:n] = xy[:, 0], xy[:, 1]
        j["coeff_off"][:n] = coff + np.arange(n, dtype=np.uint32) * nn * nn
        j["dc"][:n] = c[:, 0, 0]
        segs.append((log2, kind, sum(len(a) for a in jobs_all), n))
        jobs_all.append(j); coeffs_all.append(c.reshape(-1)); coff += c.size
        y0 += ((n + per_row - 1) // per_row) * nn
    assert y0 <= 256
    jobs = np.concatenate(jobs_all); coeffs = np.concatenate(coeffs_all)

    class Seg(C.Structure):
        _fields_ = [("log2_size", C.c_int32), ("kind", C.c_int32), ("first_job", C.c_int32), ("njobs", C.c_int32)]
    sarr = (Seg * len(segs))(*[Seg(*s) for s in segs])
    d_plane, d_jobs, d_coeffs = G.to_dev(plane), G.to_dev(jobs), G.to_dev(coeffs)
    L.check(L.load_library().ohevc_dev_tu_multi(L.planes_of([d_plane, None, None]), bd, sarr, len(segs), C.c_void_p(d_jobs.data_ptr()),
                                                C.c_void_p(d_coeffs.data_ptr()), C.c_void_p(G.stream())))
    G.sync()
    assert np.array_equal(G.to_host(d_plane, np.uint16), want)
