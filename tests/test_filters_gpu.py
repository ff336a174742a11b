"""-m gpu parity tests of deblocking and SAO vs the CPU oracle."""
import numpy as np
import pytest

from openhevc_amd import lib as L
import gpu_util as G

pytestmark = pytest.mark.gpu


def smooth_plane(rng, bd, h, w):
    """Piecewise-smooth content with steps on the 8x8 grid so off / normal / strong filtering all occur."""
    base = rng.integers(40, 200, size=(h // 8 + 1, w // 8 + 1)) << (bd - 8)
    p = np.kron(base, np.ones((8, 8), dtype=np.int64))[:h, :w]
    p = p + rng.integers(-3, 4, size=(h, w)) * (1 << (bd - 8)) // 2
    # keep some block pairs almost equal (small steps -> strong filter candidates)
    mask = rng.random((h // 8 + 1, w // 8 + 1)) < 0.5
    flat = np.kron(mask, np.ones((8, 8), dtype=np.int64))[:h, :w]
    p = np.where(flat, (p // (16 << (bd - 8))) * (16 << (bd - 8)) + rng.integers(0, 3, size=(h, w)), p)
    return np.clip(p, 0, (1 << bd) - 1).astype(G.pixdt(bd))


@pytest.mark.parametrize("bd", [8, 10, 12, 14])
def test_deblock_vertical_then_horizontal(oracle, bd):
    rng = np.random.default_rng(700 + bd)
    H, W = 96, 160
    planes = [smooth_plane(rng, bd, H, W), smooth_plane(rng, bd, H // 2, W // 2 + 16), rng.integers(0, 1 << bd, size=(H // 2, W // 2 + 16)).astype(G.pixdt(bd))]
    for vertical in (1, 0):
        jobs = []
        for pl, p in enumerate(planes):
            h, w = p.shape
            for y in range(0, h, 8):
                for x in range(0, w, 8):
                    if (vertical and x == 0) or (not vertical and y == 0) or rng.random() < 0.15:
                        continue
                    j = np.zeros(1, L.DBK_JOB)[0]
                    j["x"], j["y"], j["plane"] = x, y, pl
                    j["flags"] = (L.DBK_VERTICAL_EDGE if vertical else 0) | (int(rng.integers(0, 16)) << 1 if rng.random() < 0.2 else 0)
                    j["beta"] = int(rng.integers(0, 65))
                    j["tc"] = [int(rng.integers(0, 25)), int(rng.integers(0, 25))]
                    jobs.append(j)
        batch = np.array(jobs, dtype=L.DBK_JOB)
        want = [p.copy() for p in planes]
        for j in batch:
            f = int(j["flags"])
            no_p = [int(bool(f & L.DBK_NO_P0)), int(bool(f & L.DBK_NO_P1))]; no_q = [int(bool(f & L.DBK_NO_Q0)), int(bool(f & L.DBK_NO_Q1))]
            tc = [int(j["tc"][0]), int(j["tc"][1])]
            if j["plane"] == 0:
                oracle.deblock_luma(bd, vertical, want[0], int(j["x"]), int(j["y"]), int(j["beta"]), tc, no_p, no_q)
            else:
                oracle.deblock_chroma(bd, vertical, want[int(j["plane"])], int(j["x"]), int(j["y"]), tc, no_p, no_q)
        d = [G.to_dev(p) for p in planes]
        d_jobs = G.to_dev(batch)
        L.dev_deblock_batch(G.planes3(d), bd, d_jobs.data_ptr(), len(batch), G.stream())
        G.sync()
        changed = 0
        for pl in range(3):
            got = G.to_host(d[pl], planes[pl].dtype)
            bad = np.argwhere(got != want[pl])
            assert bad.size == 0, f"bd={bd} vertical={vertical} plane={pl}: {len(bad)} mismatches, first {bad[:3].tolist()}"
            changed += int((got != planes[pl]).sum())
            planes[pl] = got
        assert changed > 100, "test content did not trigger the filters"


@pytest.fixture(params=[0, 2, 3], ids=["shipped_wide", "lds_window", "lds_window_interior_ring_split"])
def sao_variant(request):
    """Both forms of the SAO kernel (include/ohevc_debug.h) must give the same samples."""
    lib = L.load_library()
    prev = lib.ohevc_debug_set_sao_variant(request.param)
    yield request.param
    lib.ohevc_debug_set_sao_variant(prev)


@pytest.mark.parametrize("bd", [8, 10, 12, 14])
def test_sao_band_and_edge(oracle, sao_variant, bd):
    rng = np.random.default_rng(800 + bd)
    H, W = 136, 208
    src = [np.ascontiguousarray(np.pad(rng.integers(0, 1 << bd, size=(h, w)).astype(G.pixdt(bd)), ((1, 1), (16, 16)), mode="edge"))
           for (h, w) in [(H, W), (H // 2, W // 2), (H // 2, W // 2)]]
    src[1] = (src[1] >> (bd - 3) << (bd - 4)).astype(src[1].dtype)      # coarse plane: many equal neighbours
    dst = [np.zeros_like(p) for p in src]
    jobs = []
    for pl in range(3):
        h, w = src[pl].shape[0] - 2, src[pl].shape[1] - 32
        ctb = 64 if pl == 0 else 32
        for y in range(0, h, ctb):
            for x in range(0, w, ctb):
                j = np.zeros(1, L.SAO_JOB)[0]
                j["x"], j["y"], j["w"], j["h"], j["plane"] = x + 16, y + 1, min(ctb, w - x), min(ctb, h - y), pl
                j["type"] = L.SAO_BAND if rng.random() < 0.3 else L.SAO_EDGE
                j["klass"] = int(rng.integers(0, 32)) if j["type"] == L.SAO_BAND else int(rng.integers(0, 4))
                j["borders"] = int(rng.integers(0, 16)) if rng.random() < 0.5 else 0
                j["restore"] = int(rng.random() < 0.5)
                j["edges"] = int(rng.integers(0, 256)) if rng.random() < 0.7 else 0
                ov = rng.integers(-31, 32, size=5) << (bd - 8 if bd <= 10 else 2)
                ov[0] = 0 if rng.random() < 0.8 else ov[0]
                j["offset_val"] = ov
                jobs.append(j)
    batch = np.array(jobs, dtype=L.SAO_JOB)
    want = [p.copy() for p in dst]
    for j in batch:
        pl = int(j["plane"]); args = (bd,)
        x, y, w, h = int(j["x"]), int(j["y"]), int(j["w"]), int(j["h"])
        ov = [int(v) for v in j["offset_val"]]
        if j["type"] == L.SAO_BAND:
            oracle.sao_band(bd, want[pl], src[pl], x, y, w, h, ov, int(j["klass"]))
        else:
            e = int(j["edges"]); b = int(j["borders"])
            oracle.sao_edge(bd, int(j["restore"]), want[pl], src[pl], x, y, w, h, ov, int(j["klass"]),
                            [b & 1, (b >> 1) & 1, (b >> 2) & 1, (b >> 3) & 1],
                            [e & 1, (e >> 1) & 1], [(e >> 2) & 1, (e >> 3) & 1], [(e >> 4) & 1, (e >> 5) & 1, (e >> 6) & 1, (e >> 7) & 1])
    d_src = [G.to_dev(p) for p in src]; d_dst = [G.to_dev(p) for p in dst]
    d_jobs = G.to_dev(batch)
    L.dev_sao_batch(G.planes3(d_dst), G.planes3(d_src), bd, d_jobs.data_ptr(), len(batch), G.stream())
    G.sync()
    for pl in range(3):
        got = G.to_host(d_dst[pl], dst[pl].dtype)
        bad = np.argwhere(got != want[pl])
        assert bad.size == 0, f"bd={bd} plane={pl}: {len(bad)} mismatches, first {bad[:3].tolist()}"


@pytest.mark.parametrize("variant", [0, 16], ids=["shipped", "general_loop"])
@pytest.mark.parametrize("ctb", [64, 32, 16])
@pytest.mark.parametrize("bd", [8, 10])
def test_sao_edge_interior_blocks(oracle, bd, ctb, variant):
    """Interior CTBs of a picture - edge classes, no picture border, no restored edge - take the wide kernel's short form (sao_edge_plain:
    aligned loads only, neighbours shifted in registers); every CTB has its own class and offsets, the picture's outer ring has the borders
    the decoder gives it.  Both forms must match sao_edge_filter (hevcdsp_template.c:372-567)."""
    lib = L.load_library()
    prev = lib.ohevc_debug_set_sao_variant(variant)
    try:
        rng = np.random.default_rng(4200 + bd + ctb)
        H, W = ctb * 5 + 8, ctb * 7 + 32                    # ragged last row / column
        shapes = [(H, W), (H // 2, W // 2), (H // 2, W // 2)]
        src = [np.ascontiguousarray(rng.integers(0, 1 << bd, size=sh).astype(G.pixdt(bd))) for sh in shapes]
        src[1] = (src[1] >> (bd - 3) << (bd - 3)).astype(src[1].dtype)      # coarse plane: many equal neighbours
        dst = [np.zeros_like(p) for p in src]
        jobs = []
        for pl in range(3):
            h, w = shapes[pl]
            c = ctb if pl == 0 else ctb // 2
            for y in range(0, h, c):
                for x in range(0, w, c):
                    j = np.zeros(1, L.SAO_JOB)[0]
                    j["x"], j["y"], j["w"], j["h"], j["plane"] = x, y, min(c, w - x), min(c, h - y), pl
                    j["type"], j["klass"] = L.SAO_EDGE, int(rng.integers(0, 4))
                    j["borders"] = (x == 0) * 1 + (y == 0) * 2 + (x + c >= w) * 4 + (y + c >= h) * 8
                    j["offset_val"] = [0] + [int(v) for v in rng.integers(-7, 8, size=4) << (bd - 8)]
                    jobs.append(j)
        batch = np.array(jobs, dtype=L.SAO_JOB)
        want = [p.copy() for p in dst]
        for j in batch:
            pl, b = int(j["plane"]), int(j["borders"])
            oracle.sao_edge(bd, 0, want[pl], src[pl], int(j["x"]), int(j["y"]), int(j["w"]), int(j["h"]), [int(v) for v in j["offset_val"]], int(j["klass"]),
                            [b & 1, (b >> 1) & 1, (b >> 2) & 1, (b >> 3) & 1])
        d_src = [G.to_dev(p) for p in src]; d_dst = [G.to_dev(p) for p in dst]
        d_jobs = G.to_dev(batch)
        L.dev_sao_batch(G.planes3(d_dst), G.planes3(d_src), bd, d_jobs.data_ptr(), len(batch), G.stream())
        G.sync()
        for pl in range(3):
            got = G.to_host(d_dst[pl], dst[pl].dtype)
            bad = np.argwhere(got != want[pl])
            assert bad.size == 0, f"bd={bd} ctb={ctb} plane={pl}: {len(bad)} mismatches, first {bad[:3].tolist()}"
    finally:
        lib.ohevc_debug_set_sao_variant(prev)


@pytest.mark.parametrize("exact", [1, 0])
@pytest.mark.parametrize("bd,cfi,log2_pu", [(8, 1, 2), (10, 2, 3), (8, 3, 2), (10, 1, 2)])
def test_sao_bypass_map(oracle, sao_variant, bd, cfi, log2_pu, exact):
    """SAO with the reference's is_pcm map (restore_tqb_pixels, hevc_filter.c:163-193): flagged min-PU blocks keep their
    deblocked samples, with the reference's half-CTB bound for subsampled chroma."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(900 + bd + cfi)
    hs, vs = int(cfi in (1, 2)), int(cfi == 1)
    H, W, ctb = 64 * 4, 64 * 5, 64
    shapes = [(H, W), (H >> vs, W >> hs), (H >> vs, W >> hs)]
    src = [np.ascontiguousarray(rng.integers(0, 1 << bd, size=sh).astype(G.pixdt(bd))) for sh in shapes]
    dst = [p.copy() for p in src]
    is_pcm = (rng.random((H >> log2_pu, W >> log2_pu)) < 0.3).astype(np.uint8) * 2
    jobs = []
    for pl in range(3):
        cw, ch = (ctb >> hs, ctb >> vs) if pl else (ctb, ctb)
        for cy in range(1, 3):                      # interior CTBs only: the ring every job reads stays inside the plane
            for cx in range(1, 4):
                j = np.zeros(1, L.SAO_JOB)[0]
                j["x"], j["y"], j["w"], j["h"], j["plane"] = cx * cw, cy * ch, cw, ch, pl
                j["type"] = L.SAO_BAND if rng.random() < 0.4 else L.SAO_EDGE
                j["klass"] = int(rng.integers(0, 32)) if j["type"] == L.SAO_BAND else int(rng.integers(0, 4))
                j["restore"] = int(rng.random() < 0.3)
                j["edges"] = int(rng.integers(0, 256)) if j["restore"] else 0
                j["offset_val"] = np.concatenate([[0], rng.integers(-7, 8, size=4) << (bd - 8)])
                jobs.append(j)
    batch = np.array(jobs, dtype=L.SAO_JOB)
    want = [p.copy() for p in dst]
    for j in batch:
        pl = int(j["plane"])
        x, y, w, h = int(j["x"]), int(j["y"]), int(j["w"]), int(j["h"])
        ov = [int(v) for v in j["offset_val"]]
        if j["type"] == L.SAO_BAND:
            oracle.sao_band(bd, want[pl], src[pl], x, y, w, h, ov, int(j["klass"]))
        else:
            e = int(j["edges"])
            oracle.sao_edge(bd, int(j["restore"]), want[pl], src[pl], x, y, w, h, ov, int(j["klass"]), [0, 0, 0, 0],
                            [e & 1, (e >> 1) & 1], [(e >> 2) & 1, (e >> 3) & 1], [(e >> 4) & 1, (e >> 5) & 1, (e >> 6) & 1, (e >> 7) & 1])
        po.restore_tqb_pixels(want[pl], src[pl], x << (hs if pl else 0), y << (vs if pl else 0), w, h, is_pcm, log2_pu,
                              hs if pl else 0, vs if pl else 0, bool(exact))
    d_src = [G.to_dev(p) for p in src]; d_dst = [G.to_dev(p) for p in dst]
    d_jobs = G.to_dev(batch); d_map = G.to_dev(is_pcm)
    L.dev_sao_batch_bypass(G.planes3(d_dst), G.planes3(d_src), bd, d_jobs.data_ptr(), len(batch), d_map.data_ptr(), is_pcm.shape[1],
                           log2_pu, hs, vs, exact, G.stream())
    G.sync()
    for pl in range(3):
        got = G.to_host(d_dst[pl], dst[pl].dtype)
        bad = np.argwhere(got != want[pl])
        assert bad.size == 0, f"bd={bd} cfi={cfi} plane={pl}: {len(bad)} mismatches, first {bad[:3].tolist()}"
    assert is_pcm.any() and not is_pcm.all()


@pytest.mark.parametrize("wide", [0, 1])
def test_band_sao_above_the_range_is_counted_and_wraps(oracle, wide):
    """A sample above the bit depth's range (constrained intra prediction above 8 bit leaves 0x8080 ones) sends the REFERENCE's band filter past
    its 32-entry table (hevcdsp_template.c:340-365): no reference output exists for such a stream.  The kernels wrap the band index - the
    oracle's definition - and count the event, so that the stream fuzzer can keep SAO on for those streams and compare all that never hit it."""
    import ctypes as C
    bd = 10
    lib = L.load_library()
    lib.ohevc_debug_sao_band_above_range.restype = C.c_long
    rng = np.random.default_rng(4200 + wide)
    H, W = 64, 64
    src = rng.integers(0, 1 << bd, size=(H, W)).astype(np.uint16)
    hits = [(3, 5), (17, 40), (63, 63)]
    for y, x in hits:
        src[y, x] = 0x8080
    j = np.zeros(1, L.SAO_JOB)
    j["x"], j["y"], j["w"], j["h"], j["plane"], j["type"], j["klass"] = 0, 0, W, H if wide else H - 1, 0, L.SAO_BAND, 5
    j["offset_val"] = [0, 12, -7, 30, -28]
    want = np.zeros_like(src)
    oracle.sao_band(bd, want, src, 0, 0, W, int(j["h"][0]), [0, 12, -7, 30, -28], 5)
    d_src, d_dst, d_jobs = G.to_dev(src), G.to_dev(np.zeros_like(src)), G.to_dev(j)
    G.sync()
    assert lib.ohevc_debug_sao_band_above_range(1) >= 0             # reset
    L.dev_sao_batch(G.planes3([d_dst]), G.planes3([d_src]), bd, d_jobs.data_ptr(), 1, G.stream())
    G.sync()
    got = G.to_host(d_dst, np.uint16)
    assert np.array_equal(got, want)
    n = lib.ohevc_debug_sao_band_above_range(1)
    expect = sum(1 for y, x in hits if y < int(j["h"][0]))
    assert expect <= n <= 2 * expect, n                              # the wide kernel counts per pair of samples
    assert lib.ohevc_debug_sao_band_above_range(0) == 0


def test_device_copy_kernel():
    """ohevc_dev_copy: the launch that makes the deblocked copy SAO reads (sizes from one 16-byte piece to several launches' worth of grid-stride)."""
    import ctypes as C
    rng = np.random.default_rng(77)
    lib = L.load_library()
    for nbytes in (16, 4096, 3 * 1920 * 1088 // 2 // 16 * 16, (256 * 16 * 256 + 5) * 16):
        src = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
        d_src, d_dst = G.to_dev(src), G.zeros_dev(nbytes + 16, np.uint8)
        L.check(lib.ohevc_dev_copy(C.c_void_p(d_dst.data_ptr()), C.c_void_p(d_src.data_ptr()), C.c_size_t(nbytes), C.c_void_p(G.stream())))
        G.sync()
        out = G.to_host(d_dst, np.uint8)
        assert np.array_equal(out[:nbytes], src) and not out[nbytes:].any()


def test_device_zero_kernel():
    """ohevc_dev_zero: the launch that clears the motion grid and the boundary-strength arrays at a frame end (a launch instead of hipMemsetAsync,
    which does not return while the stream waits for another stream's event)."""
    import ctypes as C
    rng = np.random.default_rng(78)
    lib = L.load_library()
    for nbytes in (16, 4096, 2 * 1920 * 1088 // 16 * 16, (256 * 16 * 256 + 5) * 16):
        d_dst = G.to_dev(rng.integers(1, 256, size=nbytes + 16, dtype=np.uint8))
        L.check(lib.ohevc_dev_zero(C.c_void_p(d_dst.data_ptr()), C.c_size_t(nbytes), C.c_void_p(G.stream())))
        G.sync()
        out = G.to_host(d_dst, np.uint8)
        assert not out[:nbytes].any() and out[nbytes:].all()
    assert lib.ohevc_dev_zero(C.c_void_p(d_dst.data_ptr() + 1), C.c_size_t(16), C.c_void_p(G.stream())) != 0, "an unaligned buffer must be refused"
