"""-m gpu parity tests of motion compensation (qpel/epel x uni/bi/weighted, edge clamping) vs the CPU oracle."""
import numpy as np
import pytest

from oracle import pyoracle as po
from openhevc_amd import lib as L
import gpu_util as G

pytestmark = pytest.mark.gpu

PAD = 80


@pytest.fixture(params=[5, 3, 6], ids=["matrix_cores", "lds_tiles", "matrix_cores_quads"])
def mc_variant(request):
    """Both forms of the motion-compensation kernel (include/ohevc_debug.h): mc4 (v_mfma_i32_16x16x32_i8, no LDS tiles; variant 5 uses it
    for the small-block entry point too - variant 4 keeps mc3 there) / mc3 (LDS tiles) / mc4 + mc4q (variant 6: the small-block entry point
    packs four blocks into one matrix-core tile)."""
    lib = L.load_library()
    prev = lib.ohevc_debug_set_mc_variant(request.param)
    yield request.param
    lib.ohevc_debug_set_mc_variant(prev)

LUMA_W = [4, 8, 12, 16, 24, 32, 48, 64]
CHROMA_W = [2, 4, 6, 8, 12, 16, 24, 32]


def expected_block(oracle, bd, luma, job, refs_padded):
    """Oracle result for one job; refs_padded[slot][plane] are edge-replicated copies (== emulated_edge_mc)."""
    pl = int(job["plane"]); w, h = int(job["w"]), int(job["h"])
    bi, wt = bool(job["flags"] & L.MC_BI), bool(job["flags"] & L.MC_WEIGHTED)
    r0 = refs_padded[int(job["ref0"])][pl]
    kw = dict(denom=int(job["denom"]), wx0=int(job["wx0"]), wx1=int(job["wx1"]), ox0=int(job["ox0"]), ox1=int(job["ox1"]))
    if not bi:
        return oracle.mc(bd, luma, po.MC_UNI_W if wt else po.MC_UNI, r0, int(job["sx0"]) + PAD, int(job["sy0"]) + PAD, w, h,
                         int(job["mx0"]), int(job["my0"]), **kw)
    tmp = oracle.mc(bd, luma, po.MC_PUT, r0, int(job["sx0"]) + PAD, int(job["sy0"]) + PAD, w, h, int(job["mx0"]), int(job["my0"]))
    src2 = np.zeros((h, 64), np.int16); src2[:, :w] = tmp
    r1 = refs_padded[int(job["ref1"])][pl]
    return oracle.mc(bd, luma, po.MC_BI_W if wt else po.MC_BI, r1, int(job["sx1"]) + PAD, int(job["sy1"]) + PAD, w, h,
                     int(job["mx1"]), int(job["my1"]), src2=src2, **kw)


def sprinkle_wild(rng, planes, bd):
    """Samples above the bit depth's range, in patches (so that clean and affected windows both occur): what the reference's
    constrained intra prediction leaves behind above 8 bit (0x8080 and interpolations of it, hevcpred_template.c:117-141)."""
    for p in planes:
        for _ in range(max(2, p.size // 2500)):
            y, x = int(rng.integers(0, p.shape[0] - 6)), int(rng.integers(0, p.shape[1] - 6))
            h, w = int(rng.integers(1, 7)), int(rng.integers(1, 7))
            p[y:y + h, x:x + w] = 0x8080 if rng.random() < 0.3 else rng.integers(1 << bd, 0x8081, size=(h, w))
        p[0, :9] = 0x8080; p[-1, -5:] = 40000          # picture corners: the clamped (edge-emulated) reads see them too


@pytest.mark.parametrize("bd,wild", [(8, 0), (10, 0), (12, 0), (14, 0), (9, 1), (10, 1), (12, 1), (14, 1)])
def test_mc_all_variants_and_edges(oracle, mc_variant, bd, wild):
    rng = np.random.default_rng(500 + bd + 50 * wild)
    W, H = 208, 144                                   # luma; chroma planes are half size (4:2:0)
    dims = [(H, W), (H // 2, W // 2), (H // 2, W // 2)]
    nslots = 3
    refs = [[rng.integers(0, 1 << bd, size=d).astype(G.pixdt(bd)) for d in dims] for _ in range(nslots)]
    if wild:
        for slot in refs:
            sprinkle_wild(rng, slot, bd)
    refs_padded = [[np.pad(p, PAD, mode="edge") for p in slot] for slot in refs]
    dst = [rng.integers(0, 1 << bd, size=d).astype(G.pixdt(bd)) for d in dims]
    # lay blocks out on a 64-sample grid so jobs never overlap in the destination
    jobs = []
    for pl in range(3):
        luma = pl == 0
        gw, gh = dims[pl][1] // 64, dims[pl][0] // 64
        cells = [(cx, cy) for cy in range(gh) for cx in range(gw)]
        for rep in range(4 if luma else 10):
            for (cx, cy) in cells:
                w = int(rng.choice(LUMA_W if luma else CHROMA_W)); h = int(rng.choice([4, 8, 12, 16, 24, 32, 64] if luma else [2, 4, 8, 12, 16, 32]))
                j = np.zeros(1, L.MC_JOB)[0]
                j["x"], j["y"], j["w"], j["h"], j["plane"] = cx * 64, cy * 64, w, h, pl
                j["flags"] = int(rng.integers(0, 4))
                far = rng.random() < 0.3               # references hanging over / far outside the picture edge
                for s in ("0", "1"):
                    lim_x, lim_y = dims[pl][1], dims[pl][0]
                    j["sx" + s] = int(rng.integers(-70, lim_x + 6)) if far else int(rng.integers(-4, lim_x - w + 4))
                    j["sy" + s] = int(rng.integers(-70, lim_y + 6)) if far else int(rng.integers(-4, lim_y - h + 4))
                    fr = 4 if luma else 8
                    j["mx" + s], j["my" + s] = int(rng.integers(0, fr)), int(rng.integers(0, fr))
                    j["ref" + s] = int(rng.integers(0, nslots))
                    j["wx" + s], j["ox" + s] = int(rng.integers(-128, 256)), int(rng.integers(-128, 128))
                j["denom"] = int(rng.integers(0, 8))
                jobs.append((rep, j))
    for rep in range(10):
        batch = np.array([j for r, j in jobs if r == rep], dtype=L.MC_JOB)
        if not len(batch):
            continue
        want = [p.copy() for p in dst]
        for j in batch:
            pl = int(j["plane"])
            want[pl][j["y"]:j["y"] + j["h"], j["x"]:j["x"] + j["w"]] = expected_block(oracle, bd, pl == 0, j, refs_padded)
        d_dst = [G.to_dev(p) for p in dst]
        d_refs = [[G.to_dev(p) for p in slot] for slot in refs]
        d_table = G.to_dev(L.planes_table(d_refs))
        d_jobs = G.to_dev(batch)
        L.dev_mc_batch(G.planes3(d_dst), d_table.data_ptr(), nslots, bd, d_jobs.data_ptr(), len(batch), G.stream())
        G.sync()
        for pl in range(3):
            got = G.to_host(d_dst[pl], dst[pl].dtype)
            bad = np.argwhere(got != want[pl])
            assert bad.size == 0, f"bd={bd} rep={rep} plane={pl}: {len(bad)} mismatches, first {bad[:3].tolist()}"


@pytest.mark.parametrize("bd,wild", [(8, 0), (10, 0), (10, 1)])
def test_mc_small_blocks_four_per_wave(oracle, mc_variant, bd, wild):
    """ohevc_dev_mc_batch_small: jobs of at most 8x8 samples, four per wavefront, incl. counts that are not multiples of 4."""
    rng = np.random.default_rng(600 + bd + 50 * wild)
    W, H = 208, 144
    dims = [(H, W), (H // 2, W // 2), (H // 2, W // 2)]
    refs = [[rng.integers(0, 1 << bd, size=d).astype(G.pixdt(bd)) for d in dims] for _ in range(2)]
    if wild:
        for slot in refs:
            sprinkle_wild(rng, slot, bd)
    refs_padded = [[np.pad(p, PAD, mode="edge") for p in slot] for slot in refs]
    dst = [rng.integers(0, 1 << bd, size=d).astype(G.pixdt(bd)) for d in dims]
    for njobs in (1, 3, 4, 157):
        jobs = []
        used = set()
        while len(jobs) < njobs:
            pl = int(rng.integers(0, 3)); luma = pl == 0
            cx, cy = int(rng.integers(0, dims[pl][1] // 8)), int(rng.integers(0, dims[pl][0] // 8))
            if (pl, cx, cy) in used:
                continue
            used.add((pl, cx, cy))
            j = np.zeros(1, L.MC_JOB)[0]
            j["x"], j["y"], j["plane"] = cx * 8, cy * 8, pl
            j["w"] = int(rng.choice([4, 8] if luma else [2, 4, 6, 8])); j["h"] = int(rng.choice([4, 8] if luma else [2, 4, 6, 8]))
            j["flags"] = int(rng.integers(0, 4))
            for s_ in ("0", "1"):
                far = rng.random() < 0.3
                j["sx" + s_] = int(rng.integers(-40, dims[pl][1] + 6)) if far else int(rng.integers(0, dims[pl][1] - 8))
                j["sy" + s_] = int(rng.integers(-40, dims[pl][0] + 6)) if far else int(rng.integers(0, dims[pl][0] - 8))
                fr = 4 if luma else 8
                j["mx" + s_], j["my" + s_] = int(rng.integers(0, fr)), int(rng.integers(0, fr))
                j["ref" + s_] = int(rng.integers(0, 2))
                j["wx" + s_], j["ox" + s_] = int(rng.integers(-128, 256)), int(rng.integers(-128, 128))
            j["denom"] = int(rng.integers(0, 8))
            jobs.append(j)
        batch = np.array(jobs, dtype=L.MC_JOB)
        want = [p.copy() for p in dst]
        for j in batch:
            pl = int(j["plane"])
            want[pl][j["y"]:j["y"] + j["h"], j["x"]:j["x"] + j["w"]] = expected_block(oracle, bd, pl == 0, j, refs_padded)
        d_dst = [G.to_dev(p) for p in dst]
        d_refs = [[G.to_dev(p) for p in slot] for slot in refs]
        d_table = G.to_dev(L.planes_table(d_refs))
        d_jobs = G.to_dev(batch)
        L.dev_mc_batch_small(G.planes3(d_dst), d_table.data_ptr(), 2, bd, d_jobs.data_ptr(), len(batch), G.stream())
        G.sync()
        for pl in range(3):
            got = G.to_host(d_dst[pl], dst[pl].dtype)
            bad = np.argwhere(got != want[pl])
            assert bad.size == 0, f"bd={bd} njobs={njobs} plane={pl}: {len(bad)} mismatches, first {bad[:3].tolist()}"
