"""-m gpu: the frame-parallel decoder over processes on a real device.  The GPU box has one GPU, so both ranks use cuda:0 and the planes
travel through a gloo group (host-staged, openhevc_amd.dist.FrameExchange._wire); everything else - slice data parsed by the owner only,
ohevc_pic_export / ohevc_pic_import on device pictures, the waits for motion fields and planes - is the path a multi-GPU run takes
with RCCL.  Every picture must equal the single-process decoder's (the committed digests)."""
import os

import pytest
import torch.multiprocessing as mp

from test_dist_cpu import build_frames_host, decoder_worker, fnv64, free_port, run_frames_hosts, write_stream_file

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(300)
def test_decoder_frame_parallel_two_processes_one_gpu():
    from oracle import pystream as ps
    assert ps.have("hip"), "oracle/_ref/libopenhevc_hip.so missing: run __graft_entry__.build() where /root/reference exists"
    names = ["ra_8b_ctb64", "ra_10b_odd", "ldb_10b", "weighted", "slices", "tiles", "cip", "fmt444_8b", "ra_14b_weighted", "ra_8b_nonref_leaves", "ra_8b_foll_leaf"]
    world, port = 2, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=decoder_worker, args=(r, world, port, names, q, "hip_device")) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=280) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for name in names:
        npics, want = res[0][name][0], res[0][name][3]
        merged = {}
        for r in range(world):
            n, digests, stats, _ = res[r][name]
            assert n == npics
            for p, dg in digests.items():
                assert p not in merged, f"{name}: picture {p} reconstructed twice"
                merged[p] = dg
        assert sorted(merged) == list(range(npics))
        assert [d for p in range(npics) for d in merged[p]] == want, f"{name}: pictures differ from the single-process decoder"
        assert sum(res[r][name][2]["awaited_planes"] for r in range(world)) > 0



@pytest.mark.timeout(300)
def test_native_transport_c_host_two_processes_one_gpu(tmp_path):
    """tests/c_host/frames_host.c on the device: two C processes (no Python, torch or gloo in them) decode one stream frame-parallel on
    GPU 0 through the native transport's sockets wire (RCCL needs one GPU per rank: that wire runs where the ranks have their own,
    bench.py --mode frames).  Expected pictures: the untouched reference decoder's, hashed here."""
    from oracle import pystream as ps
    from test_stream_cpu import load_golden
    assert ps.have("hip") and ps.have("c")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = build_frames_host(str(tmp_path))
    product = os.path.join(root, "openhevc_amd", "libohevc_hip.so")
    for k, name in enumerate(["ra_8b_ctb64", "ra_10b_odd", "ra_8b_foll_leaf", "tiles", "ra_14b_weighted"]):
        aus, _ = load_golden(name)
        want = [[fnv64(pl.tobytes()) for pl in f] for f in ps.decode_stream("c", aus)]
        sf = str(tmp_path / f"{name}.bin")
        write_stream_file(sf, aus)
        codes, errs, merged, stats = run_frames_hosts(exe, ps.lib_path("hip"), product, sf, 2, "sockets", f"127.0.0.1:{free_port() + 16 * k}", str(tmp_path))
        assert codes == [0, 0], (name, codes, errs)
        assert [merged[p] for p in range(len(want))] == want, f"{name}: pictures differ from the single-process decoder"
        assert sum(s["awaited_planes"] for s in stats) > 0 and sum(s["failed"] for s in stats) == 0


@pytest.mark.timeout(300)
@pytest.mark.parametrize("geometry", [(416, 240, 1, 8), (1920, 1080, 1, 10)], ids=["416x240_8b", "1080p_10b"])
def test_rccl_wire_executes_on_one_rank(tmp_path, geometry):
    """The RCCL wire of the native transport (openhevc_amd/csrc/frames_native.hip: hand-declared rccl.h entry points loaded with dlopen) on the
    one GPU this box has: a 1-rank communicator (the rendezvous, ncclCommInitRank behind its watchdog), then one picture through
    ohevc_frames_transport_selftest - stage, ohevc_pic_export, ONE ncclGroup of four ncclBroadcast on device memory (three planes + the
    motion field), the completion event, ohevc_pic_import - and what lands in a second picture-store slot and in the motion-field buffer
    must be what was sent.  With more GPUs the same calls move the bytes over xGMI (bench.py --mode frames --gpus N)."""
    import numpy as np
    from openhevc_amd import lib as L
    from openhevc_amd.dist import NativeFrameTransport
    w, h, cfi, bd = geometry
    rng = np.random.default_rng(77)
    dt = np.uint16 if bd > 8 else np.uint8
    shapes = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
    planes = [rng.integers(0, 1 << bd, size=s).astype(dt) for s in shapes]
    mvf = rng.integers(0, 256, size=((w + 3) // 4) * ((h + 3) // 4) * 24, dtype=np.uint8).tobytes()     # a TEST_MV_POC MvField is 24 bytes (hevc.h:1032-1041)
    ctx = L.Ctx(0)
    try:
        src = ctx.pic_alloc(w, h, cfi, bd)
        dst = ctx.pic_alloc(w, h, cfi, bd)
        ctx.pic_upload(src, planes)
        ctx.pic_upload(dst, [np.zeros_like(p) for p in planes])
        t = NativeFrameTransport(L.load_library(), 0, 1, 0, NativeFrameTransport.WIRE_RCCL, str(tmp_path / "rccl_id"), timeout_s=60)
        try:
            for _ in range(3):                      # the pools of staging buffers are reused from the second picture on
                got_mvf = t.selftest(ctx.h, src, dst, 0, mvf)
                assert got_mvf == mvf, "the motion field changed on its way through ncclBroadcast"
                got = ctx.pic_download(dst, shapes, dt)
                for c in range(3):
                    assert np.array_equal(got[c], planes[c]), f"plane {c} changed on its way through ncclBroadcast"
            t.finish()
            assert t.error is None
        finally:
            t.close()
        assert not os.path.exists(str(tmp_path / "rccl_id"))
    finally:
        ctx.close()
