"""-m gpu (and, through tests/test_hipemu_cpu.py, the CPU tier over the emulated device code): the band-chunked picture exchange of the native
transport (include/ohevc_frames.h, openhevc_amd/csrc/frames_native.hip).

The reference's frame threads publish a picture CTB row by CTB row (ff_thread_report_progress at the end of every CTB row, hevc.c:2934-2937)
and a dependent picture waits for the rows its motion vectors reach (hevc_await_progress, hevc.c:1951-1958).  Across processes the unit of both
is a band of CTU rows: the motion field travels first, then the bands; a subscriber's await_rows returns once the bands that hold the rows
it names are in its picture store.  Two ranks here are two threads of this process, each with its own transport (sockets wire: one GPU) and
its own ohevc_ctx."""
import ctypes as C
import threading

import numpy as np
import pytest

from openhevc_amd import lib as L
from openhevc_amd.dist import NativeFrameTransport, _FramesMode

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _pair(port, bands):
    """two connected transports (rank 0, rank 1); creation blocks until the peer is there, so each is made on a thread of its own"""
    lib = L.load_library()
    lib.ohevc_frames_transport_set_bands.argtypes = [C.c_void_p, C.c_int]
    out, err = [None, None], []

    def make(r):
        try:
            out[r] = NativeFrameTransport(lib, r, 2, 0, NativeFrameTransport.WIRE_SOCKETS, f"127.0.0.1:{port}", timeout_s=30)
        except Exception as e:          # noqa: BLE001
            err.append(e)
    th = [threading.Thread(target=make, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not err, err
    for t in out:
        assert lib.ohevc_frames_transport_set_bands(t.h, bands) == 0
    return out


@pytest.mark.timeout(120)
@pytest.mark.parametrize("geometry", [(416, 240, 1, 8), (200, 136, 1, 10), (256, 256, 3, 8)], ids=["416x240_420_8b", "200x136_420_10b", "256x256_444_8b"])
def test_a_subscriber_gets_the_bands_it_asks_for(geometry):
    w, h, cfi, bd = geometry
    rng = np.random.default_rng(5)
    dt = np.uint16 if bd > 8 else np.uint8
    sub = (1, 1) if cfi == 3 else (2, 2)
    shapes = [(h, w), (-(-h // sub[1]), -(-w // sub[0])), (-(-h // sub[1]), -(-w // sub[0]))]
    planes = [rng.integers(0, 1 << bd, size=s).astype(dt) for s in shapes]
    old = [np.full(s, 7, dt) for s in shapes]
    mvf = rng.integers(0, 256, size=((w + 3) // 4) * ((h + 3) // 4) * 24, dtype=np.uint8).tobytes()
    ctu_rows = (h + 63) // 64
    t0, t1 = _pair(_free_port(), 8)
    ca, cb = L.Ctx(0), L.Ctx(0)
    try:
        src, dst = ca.pic_alloc(w, h, cfi, bd), cb.pic_alloc(w, h, cfi, bd)
        ca.pic_upload(src, planes)
        cb.pic_upload(dst, old)
        m0, m1 = _FramesMode.from_address(t0.mode), _FramesMode.from_address(t1.mode)
        assert m1.await_rows, "the native transport fills await_rows"
        assert m1.subscribe(m1.user, 0, cb.h, dst, len(mvf)) == 0
        assert m0.publish(m0.user, 0, ca.h, src, mvf, len(mvf), 0) == 0, L.load_library().ohevc_last_error()
        got_mvf = C.create_string_buffer(len(mvf))
        assert m1.await_motion(m1.user, 0, got_mvf, len(mvf)) == 0
        assert got_mvf.raw == mvf

        def check(upto_band):
            got = cb.pic_download(dst, shapes, dt)
            for c in range(3):
                vs = 0 if shapes[c][0] == h else 1
                edge = min(shapes[c][0], ((upto_band + 1) * 64) >> vs) if upto_band + 1 < ctu_rows else shapes[c][0]
                assert np.array_equal(got[c][:edge], planes[c][:edge]), f"plane {c}: rows of bands 0..{upto_band} differ from what the owner sent"
                assert np.all(got[c][edge:] == 7), f"plane {c}: rows beyond band {upto_band} were touched"
        # luma row 10 lies in band 0: exactly one band goes into the store, the rest of the slot keeps its old content
        assert m1.await_rows(m1.user, 0, cb.h, dst, 10) == 0
        assert t1.stats["bands_imported"] == 1
        check(0)
        if ctu_rows >= 3:
            assert m1.await_rows(m1.user, 0, cb.h, dst, 64 + 5) == 0          # a row of band 1
            assert t1.stats["bands_imported"] == 2
            check(1)
            assert m1.await_rows(m1.user, 0, cb.h, dst, 3) == 0               # nothing new to do
            assert t1.stats["bands_imported"] == 2
        # a row of the last band completes the picture: 1 tells the caller so (the transport forgets the picture once both parts were consumed,
        # so a caller that kept asking with larger rows would be told "never subscribed")
        assert m1.await_rows(m1.user, 0, cb.h, dst, h - 1) == 1
        assert t1.stats["bands_imported"] == ctu_rows
        check(ctu_rows - 1)
        for t in (t0, t1):
            t.finish()
            assert t.error is None
            assert t.stats["wire_ranks"] == 2
        assert t0.stats["bytes"] == t1.stats["bytes"] == sum(p.shape[0] * _stride(t0, ca, src, c) for c, p in enumerate(planes)) + len(mvf) + 8
    finally:
        t0.close()
        t1.close()
        ca.close()
        cb.close()


def _stride(t, ctx, slot, plane):
    class Plane(C.Structure):
        _fields_ = [("data", C.c_void_p), ("stride", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]
    pl = (Plane * 3)()
    assert L.load_library().ohevc_pic_planes(ctx.h, slot, pl) == 0
    return pl[plane].stride


@pytest.mark.timeout(120)
def test_whole_pictures_and_bands_deliver_the_same_picture():
    """max_bands 1 = the exchange of rounds 2-4 (one message per plane); 8 = bands of CTU rows: same bytes, same picture"""
    w, h, cfi, bd = 832, 480, 1, 8
    rng = np.random.default_rng(6)
    shapes = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
    planes = [rng.integers(0, 256, size=s).astype(np.uint8) for s in shapes]
    mvf = bytes(((w + 3) // 4) * ((h + 3) // 4) * 24)
    totals = []
    for bands in (1, 8, 3):
        t0, t1 = _pair(_free_port(), bands)
        ca, cb = L.Ctx(0), L.Ctx(0)
        try:
            src, dst = ca.pic_alloc(w, h, cfi, bd), cb.pic_alloc(w, h, cfi, bd)
            ca.pic_upload(src, planes)
            cb.pic_upload(dst, [np.zeros_like(p) for p in planes])
            m0, m1 = _FramesMode.from_address(t0.mode), _FramesMode.from_address(t1.mode)
            for index in range(3):                               # the staging pools are reused from the second picture on
                assert m1.subscribe(m1.user, index, cb.h, dst, len(mvf)) == 0
                assert m0.publish(m0.user, index, ca.h, src, mvf, len(mvf), 0) == 0
                assert m1.await_planes(m1.user, index, cb.h, dst) == 0
                got = cb.pic_download(dst, shapes, np.uint8)
                assert all(np.array_equal(a, b) for a, b in zip(got, planes))
                assert m1.release(m1.user, index) == 0
            t0.finish(), t1.finish()
            assert t0.error is None and t1.error is None
            ctu_rows = (h + 63) // 64
            per_band = -(-ctu_rows // bands)
            assert t1.stats["bands_imported"] == 3 * -(-ctu_rows // per_band)
            totals.append(t1.stats["bytes"])
        finally:
            t0.close(), t1.close(), ca.close(), cb.close()
    assert len(set(totals)) == 1


def test_reach_of_the_recorded_motion_compensation():
    """ohevc_frame_ref_reach: the deepest luma row of a reference picture the open frame's prediction blocks read, taps included -
    y0 + (mv.y >> 2) + nPbH + 4 for luma (the reference waits for + 9, hevc.c:1951-1958), chroma converted to luma rows"""
    lib = L.load_library()
    ctx = L.Ctx(0)
    try:
        w, h = 416, 240
        ref_a, ref_b, cur = (ctx.pic_alloc(w, h, 1, 8) for _ in range(3))
        z = [np.zeros((h, w), np.uint8), np.zeros((h // 2, w // 2), np.uint8), np.zeros((h // 2, w // 2), np.uint8)]
        ctx.pic_upload(ref_a, z), ctx.pic_upload(ref_b, z)
        ctx.frame_begin(cur)
        assert lib.ohevc_frame_ref_reach(ctx.h, ref_a) == -1
        j = np.zeros(1, L.MC_JOB)
        j["x"], j["y"], j["w"], j["h"], j["plane"] = 32, 16, 16, 8, 0
        j["ref0"], j["sx0"], j["sy0"] = ref_a, 30, 40
        ctx.rec_mc(j)
        assert lib.ohevc_frame_ref_reach(ctx.h, ref_a) == 40 + 8 + 4
        assert lib.ohevc_frame_ref_reach(ctx.h, ref_b) == -1
        j["flags"], j["ref1"], j["sx1"], j["sy1"] = L.MC_BI, ref_b, 0, 100
        j["sy0"] = 10
        ctx.rec_mc(j)
        assert lib.ohevc_frame_ref_reach(ctx.h, ref_a) == 52          # the maximum stays
        assert lib.ohevc_frame_ref_reach(ctx.h, ref_b) == 100 + 8 + 4
        c = np.zeros(1, L.MC_JOB)                                       # a chroma block: rows in luma units
        c["x"], c["y"], c["w"], c["h"], c["plane"] = 8, 8, 8, 4, 1
        c["ref0"], c["sx0"], c["sy0"] = ref_a, 0, 60
        ctx.rec_mc(c)
        assert lib.ohevc_frame_ref_reach(ctx.h, ref_a) == ((60 + 4 + 2) << 1) + 1
        ctx.frame_reconstruct()
        ctx.frame_end()
        ctx.sync()
        ctx.frame_begin(cur)                                            # a new frame starts from nothing
        assert lib.ohevc_frame_ref_reach(ctx.h, ref_a) == -1
        ctx.frame_end()
    finally:
        ctx.close()
