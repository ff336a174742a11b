"""Stream synthesiser + bitstream-level oracle, CPU only.

The synthesiser (oracle/pystream.py + oracle/synth_gen.c) is pinned by the UNTOUCHED reference decoder: every stream
it makes must decode, on oracle/_ref/libopenhevc_c.so, to exactly the pictures the generator reconstructed.  The
committed fixtures (tests/golden/streams.npz, made by tests/golden/make_streams.py) carry the MD5 of those pictures.
"""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from oracle import pystream as ps
from stream_cases import CASES

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "streams.npz")

needs_c = pytest.mark.skipif(not ps.have("c"), reason="oracle/_ref/libopenhevc_c.so not built (needs /root/reference)")
needs_gen = pytest.mark.skipif(not (ps.have("c") and ps.have("gen")), reason="oracle/_ref decoder libraries not built")


def load_golden(name):
    z = np.load(GOLDEN)
    data = z[name + ".data"].tobytes()
    aus, o = [], 0
    for n in z[name + ".sizes"]:
        aus.append(data[o:o + int(n)])
        o += int(n)
    return aus, [str(m) for m in z[name + ".md5"]]


def frames_md5(frames):
    return [hashlib.md5(pl.tobytes()).hexdigest() for f in frames for pl in f]


@needs_gen
def test_cabac_tables_match_reference():
    """rangeTabLps / transIdxLps as restated in synth_gen.c vs the reference's packed ff_h264_cabac_tables
    (cabac.h:39-43, used as lps_range[2*(range&0xC0)+state] and mlps_state[128+state] / [127-state], cabac.c:113-121)."""
    G = ps._load("gen")
    with ps.Decoder("c"):
        tab = (C.c_uint8 * (512 + 4 * 2 * 64 + 4 * 64 + 63)).in_dll(ps._load("c"), "ff_h264_cabac_tables")
        tab = np.frombuffer(tab, dtype=np.uint8).copy()
    lps = np.ctypeslib.as_array(G.ohsyn_table_range_lps(), shape=(64, 4))
    trans = np.ctypeslib.as_array(G.ohsyn_table_trans_lps(), shape=(64,))
    for p in range(64):
        for mps in range(2):
            s = 2 * p + mps
            for q in range(4):
                assert tab[512 + q * 128 + s] == lps[p, q], (p, q)
            if p < 63:      # state 63 is only reached by the terminate path
                nxt_mps = tab[1024 + 128 + s]
                assert nxt_mps == 2 * min(p + 1, 62) + mps, p
                nxt_lps = tab[1024 + 127 - s]
                want_mps = (1 - mps) if p == 0 else mps
                assert nxt_lps == 2 * trans[p] + want_mps, p


@needs_c
@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_stream_decodes_to_recorded_md5(name):
    aus, md5 = load_golden(name)
    out = ps.decode_stream("c", aus)
    assert len(out) == CASES[name]["nframes"]
    assert frames_md5(out) == md5


@pytest.mark.skipif(not ps.have("sse"), reason="oracle/_ref/libopenhevc_sse.so not built (needs /root/reference)")
@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_sse_decoder_matches_recorded_md5(name):
    """The reference AS SHIPPED ON x86 (oracle/_ref/libopenhevc_sse.so: ARCH_X86 1, the SSE4 intrinsics of libavcodec/x86/hevcdsp_init.c:403-640
    and x86/hevcpred_init.c:31-41 wired in, deblocking forwarded to C because there is no yasm here - oracle/sse_stubs.c) is the CPU baseline
    beside every whole-decoder number: pinned, picture for picture, to what the pure-C decoder outputs on every golden stream.  Above 10 bit
    and for 4:2:2 / 4:4:4 its init leaves the C functions in place (x86/hevcdsp_init.c only knows depths 8 and 10)."""
    aus, md5 = load_golden(name)
    out = ps.decode_stream("sse", aus)
    assert len(out) == CASES[name]["nframes"]
    assert frames_md5(out) == md5


@needs_gen
@pytest.mark.parametrize("name", sorted(CASES))
def test_generator_is_deterministic_and_pinned(name):
    """Regenerating a case gives the committed bytes, and the generator's own reconstruction is what the clean decoder
    outputs (checked through the MD5s)."""
    aus, md5 = load_golden(name)
    new_aus, gen_frames = ps.generate(ps.StreamParams(**CASES[name]))
    assert [bytes(a) for a in new_aus] == aus
    assert frames_md5(gen_frames) == md5


@needs_gen
@pytest.mark.parametrize("threads,thread_type,kw", [
    (4, 2, dict(gop="lowdelay_b", nframes=5, seed=201, wpp=1, width=416, height=240)),
    (4, 2, dict(gop="lowdelay_b", nframes=5, seed=202, tiles=(3, 2), width=416, height=240)),
    (3, 1, dict(gop="random_access", nframes=17, seed=203, width=416, height=240)),
    (4, 2, dict(gop="lowdelay_b", nframes=5, seed=204, slices_per_picture=3, wpp=1, dependent_slices=1, width=416,
                height=240)),
])
def test_entry_points_hold_under_threaded_decoding(threads, thread_type, kw):
    """The WPP / tile entry points written into the slice headers are only read by the reference's threaded paths
    (hls_slice_data, hevc.c:3017-3100): decode with slice threads and frame threads and compare with one thread."""
    aus, gen_frames = ps.generate(ps.StreamParams(**kw))
    one = ps.decode_stream("c", aus)
    many = ps.decode_stream("c", aus, threads, thread_type)
    assert len(one) == len(many) == kw["nframes"]
    assert frames_md5(one) == frames_md5(many) == frames_md5(gen_frames)


def test_bit_writer_and_escaping():
    b = ps.Bits()
    b.ue(0); b.ue(1); b.ue(7); b.se(-2); b.se(3); b.u(3, 5)
    bits = "".join(map(str, b.b))
    assert bits == "1" + "010" + "0001000" + "00101" + "00110" + "101"
    assert ps.escape(bytes([0, 0, 1, 0, 0, 0, 0, 3, 5, 0, 0])) == bytes([0, 0, 3, 1, 0, 0, 3, 0, 0, 3, 3, 5, 0, 0])


def test_gop_plans_keep_every_reference_alive():
    for gop, n in (("lowdelay_p", 7), ("lowdelay_b", 7), ("random_access", 20), ("intra", 3)):
        pics = ps.plan_gop(ps.StreamParams(gop=gop, nframes=n))
        assert sorted(p.poc for p in pics) == list(range(n))
        decoded, dpb = set(), set()
        for p in pics:
            rps = {q for q, _ in p.rps_neg + p.rps_pos}
            assert rps <= dpb, (gop, p.poc)                 # only pictures still in the DPB may be named
            used = {q for q, u in p.rps_neg + p.rps_pos if u}
            if p.slice_type != ps.SLICE_I:
                assert used, (gop, p.poc)
            assert all(q < p.poc for q, _ in p.rps_neg) and all(q > p.poc for q, _ in p.rps_pos)
            dpb = rps | {p.poc}
            decoded.add(p.poc)
            assert len(dpb) <= 6


needs_hip_lib = pytest.mark.skipif(not (ps.have("c") and ps.have("hip")), reason="oracle/_ref/libopenhevc_hip.so not built")


def _record_only_counts(aus, threads, monkeypatch, thread_type=1):
    """Decode with the HIP-backed reference decoder in record-only mode (include/ohevc_debug.h: every table slot and the
    whole recorder run, no device is touched, no pixels are produced) and return the per-stream job counters."""
    monkeypatch.setenv("OHHIP_RECORD_ONLY", "1")
    L = ps._load("hip")
    sec, cnt = C.c_double(), (C.c_longlong * 8)()
    L.ohdec_backend_profile(C.byref(sec), cnt)                 # reset
    out = ps.decode_stream("hip", aus, threads, thread_type)
    L.ohdec_backend_profile(C.byref(sec), cnt)
    return len(out), dict(frames=cnt[0], tu=cnt[2], mc=cnt[3], intra=cnt[4], dbk=cnt[5], sao=cnt[6])


@needs_hip_lib
@pytest.mark.parametrize("name", ["ldb_8b", "ra_10b_odd", "pcm", "cip", "tiles", "slices_dep_wpp", "rext", "small_blocks"])
def test_recording_front_end_without_a_device(name, monkeypatch):
    """Host logic of the drop-in on real (synthetic) streams, no GPU: the recording table slots, pointer registry, job
    builders and the ctx recorder see every call the reference front-end makes; the number of recorded jobs must not
    depend on the number of frame threads (each thread records into its own context, shared picture store)."""
    aus, _ = load_golden(name)
    n1, c1 = _record_only_counts(aus, 1, monkeypatch)
    n4, c4 = _record_only_counts(aus, 4, monkeypatch)
    assert n1 == n4 == CASES[name]["nframes"]
    assert c1 == c4 and c1["frames"] == CASES[name]["nframes"]
    assert c1["tu"] > 0 and c1["intra"] > 0 and c1["dbk"] > 0
    if CASES[name]["gop"] != "intra":
        assert c1["mc"] > 0


@needs_hip_lib
@pytest.mark.parametrize("name", ["wpp", "tiles", "slices_dep_wpp", "tiles_nolf"])
def test_slice_threads_record_into_one_context(name, monkeypatch):
    """The reference's slice threads (WPP rows / tiles of one picture decoded by pool threads, hevc.c:2744-2920,3017-3100):
    every worker binds to the picture's context (ohevc_tables_set_concurrent + ohevc_tables_bind per worker) and no table
    call may be lost -- same job counts as with one thread, with slice threads and with frame+slice threads."""
    aus, _ = load_golden(name)
    n1, c1 = _record_only_counts(aus, 1, monkeypatch)
    n2, c2 = _record_only_counts(aus, 4, monkeypatch, 2)
    n4, c4 = _record_only_counts(aus, 4, monkeypatch, 4)
    assert n1 == n2 == n4 == CASES[name]["nframes"]
    assert c1 == c2 == c4


# ------------------------------------------------------------------ the whole host layer, bit-exact, without a GPU
# OHHIP_SW_EXEC=1: the HIP-backed decoder records as usual (table slots, pointer registry, job builders, dependency levels,
# filter-lag flags, bypass map, slice-thread recorders) but no device exists; at every frame end the recorded jobs are executed
# by the CPU oracle on the decoder's own frames (oracle/sw_exec.c through the frame sink of include/ohevc_debug.h).  The pictures
# must be those of the untouched decoder.  The kernels themselves are what the -m gpu tests check.
@needs_hip_lib
@pytest.mark.parametrize("bulk_filters", ["1", "0"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_recorded_jobs_executed_by_the_oracle_reproduce_the_reference(name, bulk_filters, monkeypatch):
    """bulk_filters 1 (the hooks' default): the reference's filter drivers are skipped and ohevc_tables_derive_filters records the
    deblocking / SAO jobs from the boundary-strength, QP, offset, SAO and slice / tile maps at the frame end (16x16-CTB streams
    with SAO keep the drivers: filter lag); 0: the drivers run and every edge arrives through its table slot."""
    monkeypatch.setenv("OHHIP_SW_EXEC", "1")
    monkeypatch.setenv("OHHIP_BULK_FILTERS", bulk_filters)
    aus, md5 = load_golden(name)
    assert frames_md5(ps.decode_stream("hip", aus)) == md5


@needs_hip_lib
@pytest.mark.parametrize("executor", ["3", "0"], ids=["ctb_tasks", "levels"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_both_intra_executors_record_the_same_pictures(name, executor, monkeypatch):
    """The default records both forms of the intra work and keeps the cheaper one per picture; forced here: CTB tasks (operations in
    decoding order per CTB, dependencies only where a block reads a neighbouring CTB) / dependency levels."""
    monkeypatch.setenv("OHHIP_SW_EXEC", "1")
    monkeypatch.setenv("OHHIP_LEVEL_LAUNCH", executor)
    aus, md5 = load_golden(name)
    assert frames_md5(ps.decode_stream("hip", aus)) == md5


@needs_hip_lib
@pytest.mark.parametrize("threads,thread_type,names", [
    (3, 1, ["ra_8b_ctb64", "ldb_10b", "weighted", "fmt422_10b_ra", "cross_444_8b", "small_blocks"]),       # frame threads
    (4, 2, ["wpp", "tiles", "slices_dep_wpp", "tiles_nolf"]),                                                # slice threads
    (4, 3, ["wpp", "tiles", "ra_8b_ctb64"]),                                                                 # frame x slice threads
])
def test_software_executor_in_every_thread_mode(threads, thread_type, names, monkeypatch):
    monkeypatch.setenv("OHHIP_SW_EXEC", "1")
    for name in names:
        aus, md5 = load_golden(name)
        for _ in range(2):
            assert frames_md5(ps.decode_stream("hip", aus, threads, thread_type)) == md5, (name, threads, thread_type)


@needs_gen
@pytest.mark.skipif(not ps.have("hip"), reason="oracle/_ref/libopenhevc_hip.so not built")
def test_fuzzed_streams_through_the_software_executor():
    """tools/fuzz_streams.py for a few seconds with OHHIP_SW_EXEC=1: random legal parameter sets in all thread modes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OHHIP_SW_EXEC="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_streams.py"), "12", "4243"], capture_output=True, text=True,
                       timeout=300, env=env)
    lines = r.stdout.strip().splitlines()
    assert lines, r.stderr[-2000:]
    res = json.loads(lines[-1])
    assert r.returncode == 0 and res["failed"] == 0 and res["streams"] >= 8, r.stdout[-3000:]   # (count: a floor that holds on a loaded box)


# ------------------------------------------------------------------ decoded-picture-hash SEI (the reference's only self-check)
MD5_KW = dict(gop="random_access", nframes=9, seed=611, width=416, height=240, log2_ctb=5, md5_sei=1)


@needs_gen
def test_md5_sei_is_accepted_by_the_untouched_decoder():
    """every access unit carries a suffix SEI with the MD5 of the generator's own reconstruction; with the
    `decode-checksum` option on, the reference's check (hevc_sei.c:28-45, hevc.c:4146-4162) must log "Correct MD5" for every plane
    of every picture -- one thread and frame threads -- and "Incorrect MD5" once a hash byte is flipped."""
    aus, gen_frames = ps.generate(ps.StreamParams(**MD5_KW))
    plain, _ = ps.generate(ps.StreamParams(**dict(MD5_KW, md5_sei=0)))
    assert all(a.startswith(b) and len(a) == len(b) + 4 + 2 + 2 + 1 + 48 + 1 for a, b in zip(aus, plain))   # start code, NAL header, type+size, hash_type, 3 x MD5, trailing
    for threads in (1, 3):
        out, (ok, bad) = ps.decode_stream("c", aus, threads, 1, checksum=True)
        assert frames_md5(out) == frames_md5(gen_frames)
        assert (ok, bad) == (3 * MD5_KW["nframes"], 0)
    broken = list(aus)
    broken[4] = broken[4][:-5] + bytes([broken[4][-5] ^ 0x40]) + broken[4][-4:]      # inside the Cr hash of picture 4
    _, (ok, bad) = ps.decode_stream("c", broken, checksum=True)
    assert (ok, bad) == (3 * MD5_KW["nframes"] - 1, 1)


@needs_gen
@needs_hip_lib
@pytest.mark.parametrize("threads", [1, 3])
def test_md5_check_runs_behind_the_frame_end_hook(threads, monkeypatch):
    """The hooked decoder (integration/hip_hooks.c) must end the frame -- run the recorded jobs, copy the picture back -- BEFORE
    hevc_decode_frame hashes the host planes (last row).  Here the jobs run through the software executor;
    tests/test_stream_gpu.py repeats it on the device."""
    monkeypatch.setenv("OHHIP_SW_EXEC", "1")
    aus, gen_frames = ps.generate(ps.StreamParams(**MD5_KW))
    out, (ok, bad) = ps.decode_stream("hip", aus, threads, 1, checksum=True)
    assert frames_md5(out) == frames_md5(gen_frames)
    assert (ok, bad) == (3 * MD5_KW["nframes"], 0)


@needs_gen
@needs_hip_lib
def test_stream_starting_at_a_missing_reference(monkeypatch):
    """A stream cut in front of its IDR's first dependants: the decoder synthesises the missing reference pictures
    (generate_missing_ref, hevc_refs.c:538-598: mid-grey host planes, no table call).  The hooks must send those samples to the
    device picture store (ohhip_frame_rps -> ohevc_pic_upload): the pictures predicted from them equal the untouched decoder's."""
    monkeypatch.setenv("OHHIP_SW_EXEC", "1")
    aus, _ = ps.generate(ps.StreamParams(gop="lowdelay_p", nframes=6, seed=612, width=192, height=128))
    cut = [aus[0]] + aus[3:]          # pictures 1 and 2 are lost; 3.. predict from them
    ref = ps.decode_stream("c", cut)
    assert len(ref) >= 3
    assert frames_md5(ps.decode_stream("hip", cut)) == frames_md5(ref)


@needs_c
@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("name", ["ra_8b_ctb64", "ldb_8b", "tiles", "slices_dep_wpp"])
def test_damaged_access_units_do_not_silence_the_stream(name, threads, monkeypatch):
    """Bit flips, a truncation and a run of 0xff in the slice data of two access units, then the clean stream again (it starts with an IDR
    picture) through the same decoder: no crash, no hang, whatever the damaged pass yields is equal to what the untouched decoder yields for
    the same bytes or is reported as an error, and the pictures of the clean pass are the golden ones.  (The recording slots, the frame
    life-cycle hooks and the software executor: host logic only.)"""
    if not ps.have("hip"):
        pytest.skip("oracle/_ref/libopenhevc_hip.so not built")
    import random
    monkeypatch.setenv("OHHIP_SW_EXEC", "1")
    aus, md5 = load_golden(name)
    rnd = random.Random(hash(name) & 0xffff)
    for mode in range(3):
        bad = [bytearray(a) for a in aus]
        for k in rnd.sample(range(1, len(bad)), min(2, len(bad) - 1)):
            a = bad[k]
            if mode == 0:
                for _ in range(8):
                    a[rnd.randrange(len(a) // 2, len(a))] ^= 1 << rnd.randrange(8)
            elif mode == 1:
                del a[len(a) * 2 // 3:]
            else:
                i = rnd.randrange(len(a) // 3, len(a) - 4)
                a[i:i + 4] = b"\xff\xff\xff\xff"
        seq = [bytes(a) for a in bad] + list(aus)
        results = {}
        for kind in ("c", "hip"):
            pics, errors = [], 0
            with ps.Decoder(kind, threads, 1) as d:
                for i, au in enumerate(seq):
                    r = d.L.ohdec_decode(d.h, au, len(au), i + 1)
                    if r < 0:
                        errors += 1
                    elif r:
                        pics.append(d._fetch())
                while True:
                    r = d.L.ohdec_flush(d.h)
                    if r <= 0:
                        break
                    pics.append(d._fetch())
            results[kind] = (pics, errors)
        ref_pics, _ = results["c"]
        hip_pics, hip_errors = results["hip"]
        n_clean = len(md5) // 3
        assert len(ref_pics) >= n_clean
        # the clean pass: the last n_clean pictures, from both decoders, are the golden ones
        assert frames_md5(ref_pics[-n_clean:]) == md5, f"{name} mode {mode}: the reference itself did not recover"
        assert len(hip_pics) >= n_clean and frames_md5(hip_pics[-n_clean:]) == md5, \
            f"{name} mode {mode} threads {threads}: {len(hip_pics)} pictures, {hip_errors} errors after the damaged pass"
