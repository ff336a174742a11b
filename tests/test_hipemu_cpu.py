"""The device code on the host: the -m gpu parity tests, re-run against tests/hipemu/libohevc_hip_emu.so.

That library is the UNCHANGED kernel and launcher source of openhevc_amd/csrc/ compiled for x86-64 against an emulation of the HIP
execution model (tests/hipemu/: lanes are fibers, barriers and wave collectives are scheduling points).  It checks the kernels'
arithmetic and index algebra in `pytest -m "not gpu"`, where no GPU exists; it says nothing about speed, and the parity tests proper
remain the -m gpu ones.  Test infrastructure only: nothing under openhevc_amd/ knows the emulator exists.
"""
import os
import subprocess

import pytest

import gpu_util as G

HERE = os.path.dirname(os.path.abspath(__file__))


def _build():
    r = subprocess.run(["make", "-s", "-j8", "-C", os.path.join(HERE, "hipemu")] + (["SAN=1"] if os.environ.get("HIPEMU_ASAN") == "1" else []),
                       capture_output=True, text=True)
    return r.returncode == 0 and os.path.exists(G.emulator_path()), r.stderr[-2000:]


_OK, _WHY = _build()
pytestmark = pytest.mark.skipif(not _OK, reason="kernel emulator did not build: " + _WHY)


@pytest.fixture(autouse=True)
def emulator():
    from openhevc_amd import lib as L
    prev = G.use_emulator()
    yield
    L._lib = prev


# tests that need the real device (full-size inputs generated on it, timing, spin-waits between workgroups)
SKIP = {
    "test_tu_gpu": {"test_full_size_batch_sampled_against_oracle"},
}


def _adopt(modname):
    mod = __import__(modname)
    for name, obj in vars(mod).items():
        if name in SKIP.get(modname, ()):
            continue
        if name.startswith("test_") and callable(obj):
            globals()["test_emu_" + modname[5:-4] + "_" + name[5:]] = obj
        elif hasattr(obj, "_fixture_function_marker") or type(obj).__name__ == "FixtureFunctionDefinition":
            globals()[name] = obj


for _m in os.environ.get("HIPEMU_MODULES", "test_tu_gpu test_mc_gpu test_filters_gpu test_dbk_maps_gpu test_boundary_strength_gpu test_intra_gpu test_shvc_gpu test_ctx_gpu test_tables_gpu test_frames_bands_gpu test_expand_gpu").split():
    _adopt(_m)


# ---------------------------------------------------------------- the whole decoder over the emulated device code
# oracle/_ref/libopenhevc_hipemu.so = the reference's decoder with the product's hooks (integration/hip_hooks.c), linked against the
# emulator library instead of libohevc_hip.so: parsing, recording, the ctx executor and every kernel, on the host.
def _stream_lib():
    from oracle import pystream as ps
    if not ps.have("hipemu") and ps.have("hip"):
        subprocess.run(["make", "-s", "-C", os.path.join(os.path.dirname(HERE), "oracle"), "hipemu"], capture_output=True)
    return ps if ps.have("hipemu") else None


def _golden_names():
    from stream_cases import CASES
    return sorted(CASES)


@pytest.mark.parametrize("executor", ["3", "0"], ids=["ctb_tasks", "levels"])
@pytest.mark.parametrize("name", _golden_names())
def test_emu_golden_stream(name, executor, monkeypatch):
    """Both executors of the intra-coded blocks, forced (the default picks one per picture): one CTB-task launch / launches per level."""
    from test_stream_cpu import frames_md5, load_golden
    ps = _stream_lib()
    if ps is None:
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")
    # CPU-suite time: the emulator is ~1000x slower than the device, which runs every stream in every executor (tests/test_stream_gpu.py)
    often = ("intra_8b", "intra_10b_ctb16", "ldb_8b", "ra_10b_odd", "cip", "tiles", "slices_dep_wpp", "rext", "small_blocks", "weighted",
             "fmt422_8b", "fmt444_8b", "fmt444_14b_cip_cross", "cross_444_8b", "pcm", "tqb")
    levels_too = ("ra_8b_ctb64", "weighted_p_10b", "pcm_10b", "wpp", "tiles_nolf", "slices_nolf", "dense_residual", "ra_14b_weighted", "ra_8b_nonref_leaves",
                  "rext_12b_sao_scale", "pcm_nolf_ctb16", "tqb_rext_422")
    if name not in often and not (executor == "0" and name in levels_too):
        pytest.skip("runs on the device only (CPU-suite time)")
    monkeypatch.setenv("OHHIP_LEVEL_LAUNCH", executor)
    aus, md5 = load_golden(name)
    assert frames_md5(ps.decode_stream("hipemu", aus)) == md5


@pytest.mark.parametrize("name", ["pcm", "tiles_nolf", "slices_nolf", "ra_14b_weighted"])
def test_emu_golden_stream_filters_derived_on_the_host(name, monkeypatch):
    """The job form of the deblocking (filters_host.hip, one record per edge) next to the default (maps, derived on the device)."""
    from test_stream_cpu import frames_md5, load_golden
    ps = _stream_lib()
    if ps is None:
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")
    monkeypatch.setenv("OHHIP_DEVICE_FILTERS", "0")
    aus, md5 = load_golden(name)
    assert frames_md5(ps.decode_stream("hipemu", aus)) == md5


@pytest.mark.parametrize("name", ["intra_8b", "intra_10b_ctb16", "ra_10b_odd", "cip", "tiles", "small_blocks", "fmt422_8b", "fmt444_14b_cip_cross"])
def test_emu_golden_stream_levels_in_reverse_order(name, monkeypatch):
    """The emulator runs the workgroups of a launch one after the other, in decoding order inside a dependency level - an order that hides a
    dependency the level computation missed (the device runs them concurrently).  ohevc_debug_set_reverse_levels submits every level back to front."""
    from test_stream_cpu import frames_md5, load_golden
    ps = _stream_lib()
    if ps is None:
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")
    monkeypatch.setenv("OHHIP_LEVEL_LAUNCH", "0")
    with ps.Decoder("hipemu") as d:
        product = d.product_lib()
    product.ohevc_debug_set_reverse_levels(1)
    try:
        aus, md5 = load_golden(name)
        assert frames_md5(ps.decode_stream("hipemu", aus)) == md5
    finally:
        product.ohevc_debug_set_reverse_levels(0)


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("name", ["ldb_10b", "intra_8b"])
def test_emu_golden_stream_copy_back_in_the_frame_end_hook(name, threads, monkeypatch):
    """OHHIP_DEFER_DOWNLOAD=0 (the default until round 5): the frame-end hook copies the picture back itself."""
    from test_stream_cpu import frames_md5, load_golden
    ps = _stream_lib()
    if ps is None:
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")
    monkeypatch.setenv("OHHIP_DEFER_DOWNLOAD", "0")
    aus, md5 = load_golden(name)
    assert frames_md5(ps.decode_stream("hipemu", aus, threads, 1)) == md5


@pytest.mark.parametrize("name", ["ra_10b_odd", "ldb_10b", "pcm", "intra_8b"])
def test_emu_golden_stream_pipelined_output(name, monkeypatch):
    """One decoding thread, the frame-end hook only issues the device work (OHHIP_DEFER_DOWNLOAD) and the application takes every picture
    one call late (decoder_harness.c: ohdec_set_pipelined): the device reconstructs picture k while the CPU parses picture k + 1."""
    from test_stream_cpu import frames_md5, load_golden
    ps = _stream_lib()
    if ps is None:
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")
    monkeypatch.setenv("OHHIP_DEFER_DOWNLOAD", "1")
    aus, md5 = load_golden(name)
    assert frames_md5(ps.decode_stream("hipemu", aus, pipelined=True)) == md5


@pytest.mark.parametrize("threads,thread_type,names", [
    (4, 1, ["ra_10b_odd", "ldb_10b", "weighted", "fmt444_8b"]),           # frame threads
    (4, 2, ["wpp", "tiles", "slices_dep_wpp"]),                            # slice threads
    (4, 3, ["wpp", "ra_10b_odd"]),                                            # both
])
def test_emu_golden_stream_thread_modes(threads, thread_type, names):
    from test_stream_cpu import frames_md5, load_golden
    ps = _stream_lib()
    if ps is None:
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")
    for name in names:
        aus, md5 = load_golden(name)
        assert frames_md5(ps.decode_stream("hipemu", aus, threads, thread_type)) == md5, name


def test_emu_golden_streams_with_kernel_variants():
    """The non-default kernel forms (include/ohevc_debug.h) through whole streams: same pictures."""
    import ctypes
    from test_stream_cpu import frames_md5, load_golden
    ps = _stream_lib()
    if ps is None:
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")
    lib = ctypes.CDLL(G.emulator_path())
    prev = lib.ohevc_debug_set_sao_variant(3)             # the LDS-window SAO kernel (bit 1) with its interior / ring split (bit 0)
    try:
        for name in ["ra_10b_odd", "tiles", "tqb", "pcm_nolf_ctb16", "rext_12b_sao_scale"]:
            aus, md5 = load_golden(name)
            assert frames_md5(ps.decode_stream("hipemu", aus)) == md5, name
    finally:
        lib.ohevc_debug_set_sao_variant(prev)


def test_emu_fuzzed_streams():
    """tools/fuzz_streams.py for a few seconds with FUZZ_BACKEND=hipemu: random legal parameter sets, all thread modes."""
    import json
    import sys
    from oracle import pystream as ps
    if _stream_lib() is None or not (ps.have("gen") and ps.have("c")):
        pytest.skip("generator / reference decoder / emulated decoder libraries not present")
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_streams.py"), "10", "991"], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, FUZZ_BACKEND="hipemu"))
    lines = r.stdout.strip().splitlines()
    assert lines, r.stderr[-2000:]
    res = json.loads(lines[-1])
    assert r.returncode == 0 and res["failed"] == 0 and res["streams"] >= 1, r.stdout[-3000:]


@pytest.mark.parametrize("name", ["intra_8b", "intra_10b_ctb16", "ra_8b_ctb64", "cip", "pcm", "tiles", "small_blocks", "fmt444_14b_cip_cross", "slices"])
def test_emu_golden_stream_with_an_early_flush_at_every_ctu_row(name, monkeypatch):
    """ohevc_frame_flush_intra (see tests/test_stream_gpu.py): every CTU row of a picture without inter prediction is a flush of its own."""
    from test_stream_cpu import frames_md5, load_golden
    ps = _stream_lib()
    if ps is None:
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")
    monkeypatch.setenv("OHHIP_FLUSH_INTRA_KIB", "1")
    aus, md5 = load_golden(name)
    assert frames_md5(ps.decode_stream("hipemu", aus)) == md5


@pytest.mark.parametrize("threads", [1, 3])
def test_emu_long_chain_stream_and_whole_coefficient_upload(threads):
    ps = _stream_lib()
    if ps is None:
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")
    from stream_exec import check_switches
    with ps.Decoder("hipemu") as d:
        product = d.product_lib()
    check_switches("hipemu", product, ["intra_8b", "ra_8b_ctb64", "small_blocks", "fmt444_14b_cip_cross"], threads, combos=((1, 2), (1, 1), (1, 0)))


# ---------------------------------------------------------------- decoder instances (integration/hip_backend.h), over the emulated device code
def _instances():
    ps = _stream_lib()
    if ps is None:
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")
    import instance_cases
    return instance_cases


def test_emu_two_decoders_decode_different_streams_concurrently():
    _instances().two_streams_concurrently("hipemu", ("ra_10b_odd", "ldb_10b"), threads=3)


def test_emu_two_decoders_with_different_options_in_one_process():
    _instances().two_decoders_with_different_options("hipemu")


@pytest.mark.parametrize("park_threads", [0, 2], ids=["helpers", "issuer_threads"])
def test_emu_parked_frame_ends_with_frame_threads(park_threads, monkeypatch):
    """ohhip_options.park_frames = 1 (ohevc_frame_end_deferred, include/ohevc_ctx.h; off by default - it lost its A/B runs on the device): a frame end
    whose reference pictures have not been issued yet is parked and issued by the thread that issues the last of them (or by the store's issuer
    threads).  Every picture must still be the untouched decoder's, and frames must really have been parked."""
    ps = _stream_lib()
    if ps is None:
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")
    from test_stream_cpu import frames_md5, load_golden
    monkeypatch.setenv("OHHIP_PARK_FRAMES", "1")
    if park_threads:
        monkeypatch.setenv("OHEVC_PARK_THREADS", str(park_threads))      # (read once per process: effective when this test is the first to make a store park)
    with ps.Decoder("hipemu") as d:
        product = d.product_lib()
    import ctypes
    product.ohevc_debug_parked_total.restype = ctypes.c_long
    before = product.ohevc_debug_parked_total()
    # (off by default and measured as a loss on the device: three streams keep the option alive - five streams x two thread counts took two minutes)
    for name, counts in (("ra_8b_ctb64", (4, 6)), ("ra_10b_odd", (4,)), ("weighted", (6,))):
        aus, md5 = load_golden(name)
        for threads in counts:
            assert frames_md5(ps.decode_stream("hipemu", aus, threads, 1)) == md5, f"{name} with {threads} frame threads and parked frame ends"
    assert product.ohevc_debug_parked_total() > before, "no frame end was parked: the test did not exercise ohevc_frame_end_deferred"


@pytest.mark.parametrize("own_frames", ["1", "0"], ids=["blocks_of_the_back_end", "the_decoders_allocations"])
def test_emu_frame_buffers_from_the_back_end_or_from_the_decoder(own_frames, monkeypatch):
    """ohhip_options.own_frames (integration/hip_backend.h): 1 (default) - ohhip_backend_attach installs a get_buffer2 that builds every frame out
    of page-locked blocks of the back end (recycled until the back end is freed); 0 - the decoder's own frame pool, page-locked by ohevc_host_pin.
    Same pictures either way, in every thread mode; with 1 blocks must really have been made, and none may outlive its decoder."""
    import ctypes
    from test_stream_cpu import frames_md5, load_golden
    ps = _stream_lib()
    if ps is None:
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")
    monkeypatch.setenv("OHHIP_OWN_FRAMES", own_frames)
    monkeypatch.setenv("OHHIP_PIN_FRAMES", "1")          # (opt-in since round 6: the guard in front of the decoder's pool is what this leg is about)
    L = ps._load("hipemu")
    made0, live0, made1, live1 = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong()
    L.ohhip_frame_pool_counts(ctypes.byref(made0), ctypes.byref(live0))
    for name, threads, tt in (("ra_10b_odd", 4, 1), ("wpp", 4, 2), ("ldb_10b", 4, 3)):
        aus, md5 = load_golden(name)
        # (the stream twice through one decoder instance - it starts with an IDR picture: buffers are recycled across the passes)
        assert frames_md5(ps.decode_stream("hipemu", aus * 2, threads, tt)) == list(md5) * 2, f"{name} threads {threads} type {tt} own_frames {own_frames}"
    L.ohhip_frame_pool_counts(ctypes.byref(made1), ctypes.byref(live1))
    assert live1.value == live0.value, "frame-buffer blocks outlived their decoders"
    assert (made1.value > made0.value) == (own_frames == "1")


@pytest.mark.parametrize("queue", ["1", "0"], ids=["queued_by_the_decoding_thread", "issued_by_fetch_output"])
def test_emu_copy_back_queued_at_the_frame_end_or_issued_at_fetch(queue, monkeypatch):
    """ohhip_options.queue_download (round 6): with the deferred copy-back and frame buffers of the back end's own, the decoding thread queues a
    picture's copy-back behind its device work at the frame end (ohevc_pic_download_queue) and ohhip_backend_fetch_output only waits for it;
    0: fetch_output issues the copies (round 5).  Same pictures in every thread mode, also when the application takes pictures one call late."""
    from test_stream_cpu import frames_md5, load_golden
    ps = _stream_lib()
    if ps is None:
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")
    monkeypatch.setenv("OHHIP_QUEUE_DOWNLOAD", queue)
    for name, threads, tt in (("ra_8b_ctb64", 4, 1), ("wpp", 4, 2)):
        aus, md5 = load_golden(name)
        assert frames_md5(ps.decode_stream("hipemu", aus * 2, threads, tt)) == list(md5) * 2, f"{name} threads {threads} type {tt} queue_download {queue}"
    aus, md5 = load_golden("ldb_10b")
    assert frames_md5(ps.decode_stream("hipemu", aus, pipelined=True)) == md5


def test_emu_host_blocks_of_the_library():
    """ohevc_host_alloc / ohevc_host_free (include/ohevc_ctx.h): 64-byte aligned memory the library made (page-locked on a device), freed without a
    context - also after the context is gone; a pointer the library did not make is refused, not freed."""
    import ctypes
    ps = _stream_lib()
    if ps is None:
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")
    with ps.Decoder("hipemu") as d:
        lib = d.product_lib()
    lib.ohevc_ctx_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    lib.ohevc_host_alloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    lib.ohevc_host_free.argtypes = [ctypes.c_void_p]
    lib.ohevc_ctx_destroy.argtypes = [ctypes.c_void_p]
    ctx = ctypes.c_void_p()
    assert lib.ohevc_ctx_create(ctypes.byref(ctx), 0) == 0
    blocks = []
    for n in (1, 4096, 3 << 20):
        p = ctypes.c_void_p()
        assert lib.ohevc_host_alloc(ctx, n, ctypes.byref(p)) == 0 and p.value and p.value % 64 == 0
        ctypes.memset(p, 0x5a, n)
        blocks.append(p)
    assert lib.ohevc_host_alloc(ctx, 0, ctypes.byref(ctypes.c_void_p())) != 0
    assert lib.ohevc_host_free(blocks.pop()) == 0
    lib.ohevc_ctx_destroy(ctx)
    foreign = ctypes.create_string_buffer(256)
    assert lib.ohevc_host_free(ctypes.addressof(foreign) + 128) != 0, "a pointer ohevc_host_alloc did not return was accepted"
    for p in blocks:                                   # (after their context)
        assert lib.ohevc_host_free(p) == 0
    assert lib.ohevc_host_free(None) == 0


def test_emu_options_structs_carry_their_size_first():
    _instances().options_struct_size_rule("hipemu")
    _instances().frames_mode_struct_size_rule("hipemu")


def test_emu_two_decoders_interleaved_on_one_application_thread():
    _instances().interleaved_on_one_thread("hipemu", ("intra_8b", "ra_10b_odd"))


def test_emu_decoders_opened_and_closed_leave_nothing_behind():
    _instances().open_close_many("hipemu", 12)      # (its growth check starts after ten cycles; the device tier runs 50: tests/test_stream_gpu.py)
