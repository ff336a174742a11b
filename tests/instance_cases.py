"""Two decoder instances in one process: what openHEVC's public API always creates (openHevcWrapper.c:27,47-93:
MAX_DECODERS 2, one AVCodecContext each) and what its frame threads multiply (hevc.c:4502-4513).  integration/hip_hooks.c keeps all of
its state in one ohhip_backend per instance (integration/hip_backend.h); these checks run against the device (tests/test_stream_gpu.py)
and against the emulated device code (tests/test_hipemu_cpu.py)."""
import ctypes as C
import threading

from oracle import pystream as ps
from test_stream_cpu import frames_md5, load_golden


def _decode_all(kind, aus, threads, out, key, barrier, options=None):
    try:
        with ps.Decoder(kind, threads, 1, options=options) as d:
            barrier.wait()                          # both decoders are open before either decodes: their lifetimes overlap fully
            frames = []
            for i, au in enumerate(aus):
                f = d.decode(au, i + 1)
                if f is not None:
                    frames.append(f)
            frames += d.flush()
        out[key] = frames
    except Exception as e:                          # noqa: BLE001  (reported by the caller's assert)
        out[key] = e
        try:
            barrier.abort()
        except Exception:                           # noqa: BLE001
            pass


def two_streams_concurrently(kind, names=("ra_8b_ctb64", "ldb_10b"), threads=4):
    """Two DIFFERENT golden streams (8 bit / 10 bit: different geometry, bit depth and GOP) through two decoders of one process at the same
    time, each with its own frame threads: every picture of both must be the untouched decoder's."""
    streams = [load_golden(n) for n in names]
    out, barrier = {}, threading.Barrier(2)
    ths = [threading.Thread(target=_decode_all, args=(kind, aus, threads, out, k, barrier)) for k, (aus, _) in enumerate(streams)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for k, (aus, md5) in enumerate(streams):
        assert not isinstance(out.get(k), Exception), out.get(k)
        assert frames_md5(out[k]) == md5, f"stream {names[k]} decoded beside {names[1 - k]} differs from the reference"


def two_decoders_with_different_options(kind, names=("ra_8b_ctb64", "cip"), threads=3):
    """One configuration surface: two decoders of one process, opened with DIFFERENT ohhip_options - one runs its
    intra-coded blocks as CTB tasks and derives the deblocking parameters on the host, the other takes dependency levels and derives them on
    the device - decode at the same time.  The options are per instance (ohevc_ctx_set_option on the contexts each back end makes); nothing
    process-wide is touched, the environment plays no part.  That each choice really is in force shows in the launches its decoder makes."""
    L = ps._load(kind)
    L.ohdec_backend_profile.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    streams = [load_golden(n) for n in names]
    opts = [dict(level_launch=3, device_filters=0), dict(level_launch=0, device_filters=1)]
    # (a) each set of options alone: same pictures, different launch counts (CTB tasks: one launch for all intra work; host-derived filters: more)
    launches = []
    for o in opts:
        sec, cnt = C.c_double(), (C.c_longlong * 8)()
        L.ohdec_backend_profile(C.byref(sec), cnt)
        with ps.Decoder(kind, 1, 1, options=o) as d:
            frames = [f for i, au in enumerate(streams[0][0]) if (f := d.decode(au, i + 1)) is not None]
            frames += d.flush()
        assert frames_md5(frames) == streams[0][1], o
        L.ohdec_backend_profile(C.byref(sec), cnt)
        launches.append(cnt[1])
    assert launches[0] != launches[1], f"both option sets made {launches[0]} launches: the per-instance options did not reach the contexts"
    # (b) both at once, each with its own frame threads
    out, barrier = {}, threading.Barrier(2)
    ths = [threading.Thread(target=_decode_all, args=(kind, aus, threads, out, k, barrier, opts[k])) for k, (aus, _) in enumerate(streams)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for k, (aus, md5) in enumerate(streams):
        assert not isinstance(out.get(k), Exception), out.get(k)
        assert frames_md5(out[k]) == md5, f"stream {names[k]} with options {opts[k]} beside the other decoder differs from the reference"


def interleaved_on_one_thread(kind, names=("intra_8b", "ra_10b_odd")):
    """The application thread drives two one-thread decoders in turn, access unit by access unit (what libOpenHevcDecode does with its two
    decoders, openHevcWrapper.c:103-125): the thread's table calls must follow the decoder it is in."""
    streams = [load_golden(n) for n in names]
    with ps.Decoder(kind, 1, 1) as a, ps.Decoder(kind, 1, 1) as b:
        decs, frames = (a, b), ([], [])
        for i in range(max(len(s[0]) for s in streams)):
            for k in (0, 1):
                if i < len(streams[k][0]):
                    f = decs[k].decode(streams[k][0][i], i + 1)
                    if f is not None:
                        frames[k].append(f)
        for k in (0, 1):
            frames[k].extend(decs[k].flush())
    for k in (0, 1):
        assert frames_md5(frames[k]) == streams[k][1], names[k]


def open_close_many(kind, n=50, name="ldp_8b"):
    """n decoders opened, used for a few pictures and closed one after the other: the registry of back ends is empty afterwards and the
    process has not grown (device contexts, streams, page locks, per-thread contexts all die with their instance)."""
    import os
    import resource
    L = ps._load(kind)
    L.ohhip_backend_live_count.restype = C.c_int
    aus, md5 = load_golden(name)

    def rss_kb():
        with open(f"/proc/{os.getpid()}/statm") as f:
            return int(f.read().split()[1]) * (resource.getpagesize() // 1024)

    base = None
    for i in range(n):
        out = ps.decode_stream(kind, aus[:3], 2 if i % 2 else 1, 1)
        assert len(out) >= 1
        assert L.ohhip_backend_live_count() == 0, "a closed decoder left its back end in the registry"
        if i == 9:
            base = rss_kb()                          # allocator pools and the runtime's caches have settled by now
    grown = rss_kb() - base
    assert grown < 64 * 1024, f"resident set grew by {grown} KiB over {n - 10} open/close cycles"
