"""world_size-2 gloo test (CPU) of the multi-GPU layer: unit sharding, wave planning, frame-parallel execution with
reference-picture broadcast.  'Reconstruction' is a deterministic stand-in (each picture = f(idx, its references)), so any
missing/late/mis-routed broadcast changes the result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openhevc_amd import dist as D


def fake_reconstruct(idx, refs, out):
    acc = torch.full_like(out[0], idx * 7 + 1)
    for r, planes in sorted(refs.items()):
        acc = acc * 31 + planes[0] * (r + 3)
    for k, t in enumerate(out):
        t.copy_((acc[: t.shape[0], : t.shape[1]] + k) % 65521)


def sequential(pictures, alloc):
    dpb = {}
    for p in pictures:
        planes = alloc(p.idx)
        fake_reconstruct(p.idx, {r: dpb[r] for r in p.refs}, planes)
        dpb[p.idx] = planes
    return dpb


def alloc(idx):
    return [torch.zeros(12, 16, dtype=torch.int64), torch.zeros(6, 8, dtype=torch.int64), torch.zeros(6, 8, dtype=torch.int64)]


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    D.init_from_env("gloo")
    pics = D.hierarchical_gop(3, 8)
    runner = D.FrameParallelRunner(alloc=alloc, reconstruct=fake_reconstruct)
    mine = runner.run(pics)
    want = sequential(pics, alloc)
    ok = all(torch.equal(t, w) for idx, planes in mine.items() for t, w in zip(planes, want[idx]))
    ok_dpb = all(torch.equal(t, w) for idx, planes in runner.dpb.items() for t, w in zip(planes, want[idx]))
    # kernel-level sharding: disjoint cover + max-over-ranks timing reduction as in bench.py
    lo, hi = D.shard_range(1000003, rank, world)
    t = torch.tensor([float(hi - lo)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    tmax = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    q.put((rank, ok, ok_dpb, sorted(mine), int(t.item()), float(tmax.item()), runner.broadcast_bytes))
    dist.barrier()
    dist.destroy_process_group()


def free_port(span=96):
    """a port p such that p .. p + span - 1 can all be bound right now (the sockets wire listens on port + rank; callers add small offsets per
    case: a lone free port with a busy neighbour made a rank die at start-up once in a while)"""
    for _ in range(200):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
        if p + span >= 65535:
            continue
        held = []
        try:
            for q in range(p, p + span):
                t = socket.socket(); held.append(t); t.bind(("127.0.0.1", q))
            return p
        except OSError:
            continue
        finally:
            for t in held:
                t.close()
    raise RuntimeError("no free port range")


def test_plan_and_gop_structure():
    pics = D.hierarchical_gop(2, 8)
    assert len(pics) == 17 and pics[0].refs == ()
    waves = D.plan_waves(pics)
    seen = set()
    for w in waves:                                   # every picture's references are in earlier waves
        for idx in w:
            assert all(r in seen for r in pics[idx].refs)
        seen.update(w)
    assert seen == set(range(17))
    assert sum(1 for p in pics if not p.is_reference) == 8       # the odd-POC leaves
    assert D.shard_range(10, 0, 3) == (0, 3) and D.shard_range(10, 2, 3) == (6, 10)
    cover = [D.shard_range(1 << 20, r, 8) for r in range(8)]
    assert cover[0][0] == 0 and cover[-1][1] == 1 << 20 and all(cover[i][1] == cover[i + 1][0] for i in range(7))


@pytest.mark.timeout(120)
def test_frame_parallel_two_ranks_gloo():
    world, port = 2, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    res.sort()
    owned = []
    for rank, ok, ok_dpb, mine, total, tmax, bbytes in res:
        assert ok and ok_dpb, f"rank {rank}: wrong picture contents"
        assert total == 1000003 and tmax == 2.0
        assert bbytes > 0
        owned += mine
    assert sorted(owned) == list(range(25))           # every picture reconstructed exactly once across ranks


# ---------------------------------------------------------------------------------------------------- the real decoder, frame-parallel
# integration/hip_frames.h + openhevc_amd.dist.FrameExchange: two processes decode ONE stream; each parses every slice header but only
# the slice data of the pictures it owns (decoding order, round-robin), reconstructs them (here: the recorded jobs executed by the CPU
# oracle on the decoder's host frames, OHHIP_SW_EXEC=1 - no GPU in this tier) and ships planes + motion field to the other one.
# Every picture must come out identical to the untouched single-process decoder (the committed digests).
def decoder_worker(rank, world, port, names, q, kind="hip"):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if kind == "hip":
        os.environ["OHHIP_SW_EXEC"] = "1"
    if kind == "hip_device":        # tests/test_dist_gpu.py: the real device, both ranks on cuda:0, planes host-staged through gloo
        kind = "hip"
    D.init_from_env("gloo")
    from oracle import pystream as ps
    from test_stream_cpu import frames_md5, load_golden
    result = {}
    for name in names:
        aus, md5 = load_golden(name)
        with ps.Decoder(kind) as d:
            # "hipemu": the device code emulated on the host - pictures live in "device" memory the emulator owns and cross the
            # processes through ohevc_pic_export / ohevc_pic_import, staged in CPU tensors
            ex = D.FrameExchange(d.product_lib(), device=torch.device("cpu") if kind == "hipemu" else None)
            d.frames_mode(ex.mode)
            out = []                                   # (output position, planes) of the pictures this rank reconstructed
            pos = 0
            def take(f):
                nonlocal pos
                if f is not None:
                    if d.frame_is_local():
                        out.append((pos, f))
                    pos += 1
            for i, au in enumerate(aus):
                take(d.decode(au, i + 1))
            while True:
                f = d.flush_one()
                if f is None:
                    break
                take(f)
            ex.finish()
            d.frames_mode(None)
            assert ex.error is None
        digests = {p: frames_md5([f]) for p, f in out}
        result[name] = (pos, digests, dict(ex.stats), md5)
    q.put((rank, result))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,kind", [(2, "hip"), (3, "hip"), (2, "hipemu")])
def test_decoder_frame_parallel_over_processes(world, kind):
    from oracle import pystream as ps
    if not (ps.have(kind) and os.path.exists(os.path.join(os.path.dirname(ps.__file__), "libohsw.so"))):
        pytest.skip("GPU-backed decoder / software executor / emulator build not present (needs the reference tree once)")
    names = ["ra_8b_ctb64", "ldb_10b", "weighted", "ra_10b_odd", "intra_8b", "slices", "tiles", "cip", "fmt444_8b", "ra_8b_nonref_leaves", "ra_8b_foll_leaf"]
    if kind == "hipemu":                # the emulated device code is slow: the streams that exercise export / import, skipping and reordering
        names = ["ra_10b_odd", "weighted", "tiles", "ra_8b_nonref_leaves", "ra_8b_foll_leaf"]
    port = free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=decoder_worker, args=(r, world, port, names, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=280) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for name in names:
        npics = res[0][name][0]
        want = res[0][name][3]
        per_pic = len(want) // npics                   # 3 digests (planes) per picture
        merged = {}
        for r in range(world):
            n, digests, stats, _ = res[r][name]
            assert n == npics
            for p, dg in digests.items():
                assert p not in merged, f"{name}: picture {p} reconstructed twice"
                merged[p] = dg
        assert sorted(merged) == list(range(npics)), f"{name}: pictures {sorted(set(range(npics)) - set(merged))} reconstructed by nobody"
        got = [d for p in range(npics) for d in merged[p]]
        assert got == want, f"{name}: pictures differ from the single-process decoder"
        assert per_pic == 3
        if name in ("ra_8b_nonref_leaves", "ra_8b_foll_leaf"):           # the four leaves of the GOP are nobody's reference: not exchanged
            assert sum(res[r][name][2]["published"] for r in range(world)) == npics - 4, res[0][name][2]
        if npics > 2 and name != "intra_8b":
            assert sum(res[r][name][2]["awaited_planes"] for r in range(world)) > 0, f"{name}: no reference picture ever crossed processes"


# A picture its owner cannot complete is published all the same, marked failed (hip_frames.h): the other process's wait for it fails at
# once - no hang until the process group's timeout - and nothing is left in flight when the exchange is torn down.
def failing_worker(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), OHHIP_SW_EXEC="1",
                      OHHIP_TEST_FAIL_INDEX="1", OHEVC_DIST_TIMEOUT_SECONDS="60")
    D.init_from_env("gloo")
    from oracle import pystream as ps
    from test_stream_cpu import load_golden
    aus, _ = load_golden("ra_8b_ctb64")
    import time
    t0 = time.time()
    failed_at = None
    with ps.Decoder("hip") as d:
        ex = D.FrameExchange(d.product_lib())
        d.frames_mode(ex.mode)
        for i, au in enumerate(aus):
            if d.L.ohdec_decode(d.h, au, len(au), i + 1) < 0:
                failed_at = i
                break
        ex.finish()                                 # must return: every collective issued has a partner
        d.frames_mode(None)
    q.put((rank, failed_at, dict(ex.stats), time.time() - t0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_owner_failure_reaches_the_other_process():
    from oracle import pystream as ps
    if not (ps.have("hip") and os.path.exists(os.path.join(os.path.dirname(ps.__file__), "libohsw.so"))):
        pytest.skip("GPU-backed decoder / software executor build not present (needs the reference tree once)")
    port = free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (f, st, dt) for r, f, st, dt in (q.get(timeout=100) for _ in range(2))}
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res[1][0] == 1 and res[1][1]["failed"] == 1          # picture 1 (decoding order) is rank 1's: its frame end reports the failure ...
    assert res[0][0] is not None and res[0][0] >= 2              # ... and rank 0 fails on the first picture that predicts from it
    assert res[0][2] < 30 and res[1][2] < 30                     # at once, not after a timeout


# ---------------------------------------------------------------------------------------------------- the native transport, driven from C
# include/ohevc_frames.h: the transport in C inside the product library, tests/c_host/frames_host.c: a C host of the frame-parallel
# decoder - no Python, torch or gloo in the processes that decode.  Here: the emulated device code and the sockets wire (the RCCL wire
# needs one GPU per rank; the protocol above the wire is the same code).
def fnv64(b):
    h = 1469598103934665603
    for v in np.frombuffer(b, dtype=np.uint8).tolist():
        h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def build_frames_host(tmpdir):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(tmpdir, "frames_host")
    subprocess.run(["gcc", "-O1", "-Wall", os.path.join(root, "tests", "c_host", "frames_host.c"), "-I" + os.path.join(root, "include"), "-ldl", "-o", exe], check=True)
    return exe


def write_stream_file(path, aus):
    with open(path, "wb") as f:
        f.write(np.uint32(len(aus)).tobytes())
        f.write(np.array([len(a) for a in aus], np.uint32).tobytes())
        for a in aus:
            f.write(a)


def run_frames_hosts(exe, decoder_so, product_so, stream_file, world, wire, rendezvous, tmpdir, device=0, env=None, bands=None):
    import subprocess
    procs, outs = [], []
    for r in range(world):
        out = os.path.join(tmpdir, f"out_{os.path.basename(stream_file)}_{r}.txt")
        outs.append(out)
        procs.append(subprocess.Popen([exe, decoder_so, product_so, stream_file, str(r), str(world), wire, rendezvous, str(device), out] + ([str(bands)] if bands else []),
                                      env=dict(os.environ, **(env or {})), stderr=subprocess.PIPE))
    errs = [p.communicate(timeout=240)[1].decode(errors="replace") for p in procs]
    codes = [p.returncode for p in procs]
    merged, stats = {}, []
    for r, out in enumerate(outs):
        assert os.path.exists(out), f"rank {r} left no output (exit code {codes[r]}): {errs[r][-1500:]}"
        for line in open(out):
            w = line.split()
            if w[0] == "stats":
                stats.append({w[i]: int(w[i + 1]) for i in range(1, len(w), 2)})
            else:
                assert int(w[0]) not in merged, f"picture {w[0]} reconstructed twice"
                merged[int(w[0])] = [int(v, 16) for v in w[1:]]
    return codes, errs, merged, stats


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,bands", [(2, None), (3, None), (2, 1)], ids=["2", "3", "2_whole_pictures"])
def test_native_transport_c_host_over_sockets(world, bands, tmp_path):
    from oracle import pystream as ps
    from test_stream_cpu import load_golden
    if not (ps.have("hipemu") and ps.have("c")):
        pytest.skip("emulator-backed decoder not built (needs the reference tree once)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = build_frames_host(str(tmp_path))
    product = os.path.join(root, "tests", "hipemu", "libohevc_hip_emu.so")
    for k, name in enumerate(["ra_10b_odd", "ra_8b_foll_leaf", "weighted"]):
        aus, _ = load_golden(name)
        want = [[fnv64(pl.tobytes()) for pl in f] for f in ps.decode_stream("c", aus)]
        sf = str(tmp_path / f"{name}.bin")
        write_stream_file(sf, aus)
        codes, errs, merged, stats = run_frames_hosts(exe, ps.lib_path("hipemu"), product, sf, world, "sockets", f"127.0.0.1:{free_port() + 16 * k}", str(tmp_path), bands=bands)
        assert codes == [0] * world, (name, codes, errs)
        assert sorted(merged) == list(range(len(want))), (name, sorted(merged))
        assert [merged[p] for p in range(len(want))] == want, f"{name}: pictures differ from the single-process decoder"
        assert all(s["pictures"] == len(want) for s in stats)
        assert sum(s["awaited_planes"] for s in stats) > 0 and sum(s["failed"] for s in stats) == 0
        assert all(s["wire_ranks"] == world for s in stats)
        # pictures cross the wire in bands of CTU rows (default) or whole (bands 1): with bands, more of them are imported than pictures waited for
        if bands == 1:
            assert sum(s["bands_imported"] for s in stats) == sum(s["awaited_planes"] for s in stats)
        else:
            assert sum(s["bands_imported"] for s in stats) > sum(s["awaited_planes"] for s in stats)


def test_rccl_rendezvous_never_accepts_a_stale_id(tmp_path):
    """The file through which rank 0 hands its ncclUniqueId to the other ranks (frames_native.hip) - ncclCommInitRank hangs without a timeout
    when a rank brings another id, so a file left behind by a crashed run must never be taken for this run's.  Three ranks (threads of this
    process; the handshake is plain file I/O) meet through a path where an earlier run left an id file, a ready file and an ack file: every
    rank must come out with THIS run's id, whichever rank starts first, and the path must be clean afterwards."""
    import ctypes as C
    import threading
    import time
    from openhevc_amd import lib as L
    lib = L.load_library()
    lib.ohevc_debug_frames_rendezvous.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
    path = str(tmp_path / "id")
    world = 3
    for order in ([0, 1, 2], [2, 1, 0], [1, 0, 2]):
        # debris of an earlier run: a complete, well-formed id file with other nonces, and the files of its rank 1
        stale = bytes([0x76, 0x72, 0x68, 0x6f, 0, 0, 0, 0]) + bytes(16) + bytes(16 * world) + bytes([0xAA] * 128)
        with open(path, "wb") as f:
            f.write(stale)
        with open(path + ".ready.1", "wb") as f:
            f.write(bytes([7] * 16))
        with open(path + ".ack.1", "wb") as f:
            f.write(bytes([9] * 16))
        want = bytes((37 * k + order[0]) & 0xff for k in range(128))
        got, rcs = {}, {}

        def run(rank):
            buf = C.create_string_buffer(want if rank == 0 else bytes(128), 128)
            rcs[rank] = lib.ohevc_debug_frames_rendezvous(path.encode(), rank, world, 30, buf)
            got[rank] = buf.raw

        ths = []
        for rank in order:
            ths.append(threading.Thread(target=run, args=(rank,)))
            ths[-1].start()
            time.sleep(0.15)                         # the ranks arrive one after the other: late rank 0, late readers
        for t in ths:
            t.join()
        assert rcs == {0: 0, 1: 0, 2: 0}, (order, rcs, lib.ohevc_last_error())
        assert all(got[r] == want for r in range(world)), f"order {order}: a rank came out with another id"
        assert sorted(os.listdir(tmp_path)) == [], sorted(os.listdir(tmp_path))


def test_rccl_rendezvous_times_out_instead_of_hanging(tmp_path):
    import ctypes as C
    from openhevc_amd import lib as L
    lib = L.load_library()
    lib.ohevc_debug_frames_rendezvous.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
    lib.ohevc_last_error.restype = C.c_char_p
    path = str(tmp_path / "id")
    with open(path, "wb") as f:                      # only a stale id file is there: rank 1 must not take it
        f.write(bytes([0x76, 0x72, 0x68, 0x6f, 0, 0, 0, 0]) + bytes(16 + 32 + 128))
    buf = C.create_string_buffer(128)
    assert lib.ohevc_debug_frames_rendezvous(path.encode(), 1, 2, 1, buf) != 0
    assert b"never published" in lib.ohevc_last_error()
    assert lib.ohevc_debug_frames_rendezvous(path.encode(), 0, 2, 1, buf) != 0      # ... and rank 0 gives up when nobody answers
    assert b"not every rank" in lib.ohevc_last_error()


# ---------------------------------------------------------------- ownership per IDR segment (ohhip_frames_mode.segment_ownership)
def segment_worker(rank, world, port, names, repeat, q, kind):
    """every stream `repeat` times through one decoder (each repetition opens with an IDR picture = one segment), frame-parallel with
    ownership per IDR segment over the native transport's sockets wire; reports, per stream, which output positions this rank reconstructed,
    their digests, and the transport's counters"""
    import zlib
    from oracle import pystream as ps
    from openhevc_amd import dist as D
    from test_stream_cpu import load_golden
    res = {}
    try:
        for k, name in enumerate(names):
            aus, _ = load_golden(name)
            got = {}
            with ps.Decoder(kind) as d:
                ex = D.NativeFrameTransport(d.product_lib(), rank, world, 0, D.NativeFrameTransport.WIRE_SOCKETS, f"127.0.0.1:{port + 16 * k}", timeout_s=60)
                ex.set_ownership(True)
                d.frames_mode(ex.mode)
                n = 0

                def took(pic):
                    nonlocal n
                    if d.frame_is_local():
                        got[n] = [zlib.crc32(pl.tobytes()) for pl in pic]
                    n += 1
                for r in range(repeat):
                    for i, au in enumerate(aus):
                        pic = d.decode(au, r * 1000 + i + 1)
                        if pic is not None:
                            took(pic)
                while True:
                    pic = d.flush_one()
                    if pic is None:
                        break
                    took(pic)
                ex.finish()
                d.frames_mode(None)
                st = ex.stats
                err = ex.error
                ex.close()
            res[name] = (n, got, st, None if err is None else str(err))
        q.put((rank, res))
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put((rank, {"error": traceback.format_exc() + str(e)}))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_idr_segments_are_decoded_by_one_rank_each_and_nothing_is_exchanged(world):
    """An IDR picture empties the decoded picture buffer: a segment needs nothing from before it.  With ohhip_frames_mode.segment_ownership the
    ranks take whole segments in turn (segment number % world), every picture is reconstructed exactly once, equals the single-process
    decoder's, and the wire carries nothing."""
    import zlib
    from oracle import pystream as ps
    from test_stream_cpu import load_golden
    if not ps.have("hipemu"):
        pytest.skip("emulator-backed decoder not built (needs the reference tree once)")
    names, repeat = ["ra_8b_ctb64", "ldb_10b"], 3
    port = free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=segment_worker, args=(r, world, port, names, repeat, q, "hipemu")) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=280) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        assert "error" not in res[r], res[r]["error"]
    for name in names:
        aus, _ = load_golden(name)
        one = [[zlib.crc32(pl.tobytes()) for pl in f] for f in ps.decode_stream("hipemu", aus)]
        per = len(one)
        merged = {}
        for r in range(world):
            n, got, st, err = res[r][name]
            assert err is None and n == per * repeat
            assert st["published"] == 0 and st["subscribed"] == 0 and st["bytes"] == 0, f"{name}: segment ownership exchanged pictures: {st}"
            for pos, dg in got.items():
                assert pos not in merged, f"{name}: picture {pos} reconstructed twice"
                merged[pos] = dg
                assert (pos // per) % world == r, f"{name}: output position {pos} (segment {pos // per}) reconstructed by rank {r}"
        assert sorted(merged) == list(range(per * repeat)), f"{name}: pictures nobody reconstructed"
        assert all(merged[pos] == one[pos % per] for pos in merged), f"{name}: pictures differ from the single-process decoder"
