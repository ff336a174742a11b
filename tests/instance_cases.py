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


# ---------------------------------------------------------------- the options structs' ABI rule (struct_size first)
class _Options(C.Structure):
    """ohhip_options as integration/hip_backend.h has it TODAY"""
    _fields_ = [("struct_size", C.c_size_t), ("device", C.c_int), ("bulk_filters", C.c_int), ("defer_download", C.c_int), ("pin_frames", C.c_int),
                ("async_issue", C.c_int), ("record_only", C.c_int), ("test_fail_index", C.c_int), ("flush_intra_kib", C.c_int),
                ("level_launch", C.c_int), ("device_filters", C.c_int), ("crash_backtrace", C.c_int), ("trace_path", C.c_char_p), ("base_layer", C.c_void_p), ("park_frames", C.c_int), ("own_frames", C.c_int), ("queue_download", C.c_int)]


class _OldOptions(C.Structure):
    """a host compiled against an OLDER header: the struct ended after flush_intra_kib"""
    _fields_ = _Options._fields_[:9]


def options_struct_size_rule(kind):
    """ohhip_backend_new reads no more of the caller's struct than struct_size says, fills what is missing with the library's defaults
    (-1 = "library default", not 0 = "host-side derivation"), and refuses a struct nobody initialised or one from a newer host."""
    L = ps._load(kind)
    L.ohhip_backend_new.restype = C.c_void_p
    L.ohhip_backend_new.argtypes = [C.c_void_p]
    L.ohhip_backend_free.argtypes = [C.c_void_p]
    L.ohhip_backend_options.argtypes = [C.c_void_p, C.c_void_p]
    L.ohhip_options_default.argtypes = [C.c_void_p]
    L.ohhip_options_size.restype = C.c_size_t
    assert L.ohhip_options_size() == C.sizeof(_Options)
    dflt = _Options()
    L.ohhip_options_default(C.byref(dflt))
    assert dflt.struct_size == C.sizeof(_Options) and dflt.level_launch == -1 and dflt.device_filters == -1 and dflt.defer_download == 1

    def effective(be):
        got = _Options()
        got.struct_size = C.sizeof(_Options)
        assert L.ohhip_backend_options(be, C.byref(got)) == 0
        return got

    # (a) zero-initialised struct (struct_size 0): refused - every field would have read as 0
    assert not L.ohhip_backend_new(C.byref(_Options()))
    # (b) a struct from a newer host (larger than the library's): refused
    big = _Options()
    L.ohhip_options_default(C.byref(big))
    big.struct_size = C.sizeof(_Options) + 8
    assert not L.ohhip_backend_new(C.byref(big))
    # (c) an older host: 9 fields, followed in memory by garbage the library must not read
    class Padded(C.Structure):
        _fields_ = [("o", _OldOptions), ("junk", C.c_ubyte * 64)]
    old = Padded()
    C.memset(C.byref(old), 0xA5, C.sizeof(old))
    old.o.struct_size = C.sizeof(_OldOptions)
    old.o.device, old.o.bulk_filters, old.o.defer_download, old.o.pin_frames = dflt.device, 1, 0, 1
    old.o.async_issue, old.o.record_only, old.o.test_fail_index, old.o.flush_intra_kib = 0, dflt.record_only, -1, 777
    be = L.ohhip_backend_new(C.byref(old))
    assert be
    got = effective(be)
    assert (got.defer_download, got.flush_intra_kib) == (0, 777)                       # what the old host said
    assert (got.level_launch, got.device_filters, got.crash_backtrace) == (dflt.level_launch, dflt.device_filters, dflt.crash_backtrace)   # defaults, not 0xA5A5A5A5
    assert got.trace_path == dflt.trace_path and got.base_layer == dflt.base_layer
    # the getter honours the asker's size too
    small = Padded()
    C.memset(C.byref(small), 0x5A, C.sizeof(small))
    small.o.struct_size = C.sizeof(_OldOptions)
    assert L.ohhip_backend_options(be, C.byref(small)) == 0
    assert small.o.flush_intra_kib == 777 and bytes(small.junk) == b"\x5a" * 64
    L.ohhip_backend_free(be)
    # (d) today's struct, one field changed
    cur = _Options()
    L.ohhip_options_default(C.byref(cur))
    cur.level_launch = 3
    be = L.ohhip_backend_new(C.byref(cur))
    assert be and effective(be).level_launch == 3
    L.ohhip_backend_free(be)


def frames_mode_struct_size_rule(kind):
    """ohhip_backend_frames_mode: struct_size first; a caller without the trailing fields (await_rows, segment_ownership) gets NULL / 0 for them,
    never the bytes behind its struct; 0 and too-large sizes are refused."""
    from openhevc_amd import dist as D
    FM = D._FramesMode
    L = ps._load(kind)
    L.ohhip_backend_new.restype = C.c_void_p
    L.ohhip_backend_new.argtypes = [C.c_void_p]
    L.ohhip_backend_free.argtypes = [C.c_void_p]
    L.ohhip_backend_frames_mode.argtypes = [C.c_void_p, C.c_void_p]
    be = L.ohhip_backend_new(None)
    assert be
    cbs = (FM.PUBLISH(lambda *a: 0), FM.SUBSCRIBE(lambda *a: 0), FM.AWAIT_MOTION(lambda *a: 0), FM.AWAIT_PLANES(lambda *a: 0), FM.RELEASE(lambda *a: 0))

    class Padded(C.Structure):
        _fields_ = [("m", FM), ("junk", C.c_ubyte * 32)]
    p = Padded()
    p.m = FM(C.sizeof(FM), 0, 2, None, *cbs)
    assert L.ohhip_backend_frames_mode(be, C.byref(p)) == 0
    p.m.struct_size = 0
    assert L.ohhip_backend_frames_mode(be, C.byref(p)) != 0
    p.m.struct_size = C.sizeof(FM) + 16
    assert L.ohhip_backend_frames_mode(be, C.byref(p)) != 0
    # an older caller: the struct ends after `release`; what follows in ITS memory is not ours to read (here: a poisoned pointer and flag)
    C.memset(C.addressof(p) + FM.await_rows.offset, 0xEE, C.sizeof(p) - FM.await_rows.offset)
    p.m.struct_size = FM.await_rows.offset
    assert L.ohhip_backend_frames_mode(be, C.byref(p)) == 0
    # missing a REQUIRED callback: refused
    p.m.struct_size = FM.await_planes.offset
    assert L.ohhip_backend_frames_mode(be, C.byref(p)) != 0
    assert L.ohhip_backend_frames_mode(be, None) == 0
    L.ohhip_backend_free(be)
