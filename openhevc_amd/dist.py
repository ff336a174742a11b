"""Multi-GPU orchestration (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The reference has no distributed layer: its frame threads share the DPB in host RAM and wait on row-progress counters
(pthread_frame.c:479-513, hevc.c:1951-1958).  The MI355X-native equivalent implemented here:

* kernel level  -- units (blocks / PUs / edges) of one batch are independent: `shard_range` splits them by rank, no
                   collective on the data path (bench.py --gpus N, "scaling": "weak").
* frame level   -- pictures are independent given their reference pictures.  `plan_waves` turns the decode-order
                   dependency DAG of a GOP into waves of mutually independent pictures; inside a wave pictures are
                   dealt round-robin to ranks; after a wave every picture that later pictures reference is broadcast
                   from its owner to all ranks, so each GPU keeps a full replica of the DPB it needs in its own HBM.
                   The collective is a broadcast per plane (direct fan-out over the point-to-point xGMI links; no
                   reduction is ever needed).  Non-reference pictures are never sent.

Nothing here touches pixels: reconstruction is injected as a callable, so the protocol is testable with the gloo backend
on CPU tensors (tests/test_dist_cpu.py) and runs unchanged on RCCL with device tensors.
"""
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend=None, timeout_s=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (set by torch.distributed.run). Returns (rank, world).
    The process group gets a timeout (OHEVC_DIST_TIMEOUT_SECONDS, default 120 s): a peer that died must not hang the others for
    gloo's 30 minutes (RCCL: until the watchdog fires)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        import datetime
        if timeout_s is None:
            timeout_s = float(os.environ.get("OHEVC_DIST_TIMEOUT_SECONDS", "120"))
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return rank, world


def shard_range(n_units: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of n_units independent units owned by `rank` (sizes differ by at most one)."""
    lo = n_units * rank // world
    hi = n_units * (rank + 1) // world
    return lo, hi


@dataclass
class Picture:
    """One picture of a decode-order stream: `refs` are the decode-order indices of the pictures it predicts from."""
    idx: int
    refs: Sequence[int] = ()
    is_reference: bool = True          # used as a reference by a later picture (else it is never broadcast)


def plan_waves(pictures: Sequence[Picture]) -> List[List[int]]:
    """Dependency levels of the picture DAG: wave k holds every picture whose references all lie in waves < k."""
    level: Dict[int, int] = {}
    for p in pictures:                 # decode order is a topological order
        level[p.idx] = 1 + max((level[r] for r in p.refs), default=-1)
    waves: List[List[int]] = [[] for _ in range(1 + max(level.values(), default=-1))]
    for p in pictures:
        waves[level[p.idx]].append(p.idx)
    return waves


def owner_of(wave: Sequence[int], world: int) -> Dict[int, int]:
    """Round-robin ownership inside one wave."""
    return {idx: k % world for k, idx in enumerate(wave)}


def hierarchical_gop(n_gops: int, gop: int = 8) -> List[Picture]:
    """Random-access style stream: I0, then per GOP the anchor P(gop) and a dyadic B hierarchy (leaves non-reference)."""
    pics: List[Picture] = [Picture(0, ())]
    poc_to_idx = {0: 0}

    def add(poc, ref_pocs, is_ref):
        idx = len(pics)
        pics.append(Picture(idx, tuple(poc_to_idx[r] for r in ref_pocs), is_ref))
        poc_to_idx[poc] = idx

    for g in range(n_gops):
        base = g * gop
        add(base + gop, (base,), True)

        def split(lo, hi):
            if hi - lo < 2:
                return
            mid = (lo + hi) // 2
            add(mid, (lo, hi), (mid - lo) > 1)
            split(lo, mid)
            split(mid, hi)

        split(base, base + gop)
    return pics


@dataclass
class FrameParallelRunner:
    """Runs a picture stream frame-parallel over the ranks of the default process group.

    alloc(idx)                  -> list of plane tensors for picture idx (device or CPU), same shapes on every rank
    reconstruct(idx, refs, out) -> fills `out` (the picture's planes) given {ref idx: planes}
    """
    alloc: Callable[[int], List[torch.Tensor]]
    reconstruct: Callable[[int, Dict[int, List[torch.Tensor]], List[torch.Tensor]], None]
    dpb: Dict[int, List[torch.Tensor]] = field(default_factory=dict)
    broadcast_bytes: int = 0

    def run(self, pictures: Sequence[Picture]):
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        by_idx = {p.idx: p for p in pictures}
        mine: Dict[int, List[torch.Tensor]] = {}
        for wave in plan_waves(pictures):
            owners = owner_of(wave, world)
            for idx in wave:                                   # 1. every rank reconstructs the pictures it owns
                if owners[idx] == rank:
                    planes = self.alloc(idx)
                    self.reconstruct(idx, {r: self.dpb[r] for r in by_idx[idx].refs}, planes)
                    mine[idx] = planes
                    if by_idx[idx].is_reference:
                        self.dpb[idx] = planes
            handles = []
            for idx in wave:                                   # 2. reference pictures go to everyone (async, all planes in flight)
                if not by_idx[idx].is_reference:
                    continue
                if owners[idx] != rank:
                    self.dpb[idx] = self.alloc(idx)
                if world > 1:
                    for t in self.dpb[idx]:
                        handles.append(dist.broadcast(t, src=owners[idx], async_op=True))
                        self.broadcast_bytes += t.numel() * t.element_size()
            for h in handles:
                h.wait()
        return mine


# ---------------------------------------------------------------------------------------------------- the real decoder, frame-parallel
# integration/hip_frames.h: every process parses the slice headers of the whole stream, the owner of a picture (decoding-order index
# % world) parses its slice data and reconstructs it on its GPU, and what later pictures need from it - the sample planes and the
# motion field - travels to the other processes.  FrameExchange is that transport: the four callbacks of ohhip_frames_mode on top of
# torch.distributed (planes: broadcast on the default group = RCCL over xGMI with device tensors; motion fields, host memory: a gloo
# group).  Collectives are issued in decoding order on every rank (publish on the owner, subscribe elsewhere), asynchronously;
# a rank blocks only in await_motion / await_planes, the two waits the reference's frame threads have (hevc_mvs.c; hevc.c:1951-1958).
import ctypes as _C

import numpy as _np


class _FramesMode(_C.Structure):
    """ohhip_frames_mode (integration/hip_frames.h)"""
    PUBLISH = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_int, _C.c_void_p, _C.c_int, _C.c_void_p, _C.c_size_t, _C.c_int)
    SUBSCRIBE = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_int, _C.c_void_p, _C.c_int, _C.c_size_t)
    AWAIT_MOTION = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_int, _C.c_void_p, _C.c_size_t)
    AWAIT_PLANES = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_int, _C.c_void_p, _C.c_int)
    RELEASE = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_int)
    AWAIT_ROWS = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_int, _C.c_void_p, _C.c_int, _C.c_int)
    _fields_ = [("struct_size", _C.c_size_t),    # first: sizeof(this struct) as THIS caller knows it (ohhip_backend_frames_mode checks it)
                ("rank", _C.c_int), ("world", _C.c_int), ("user", _C.c_void_p), ("publish", PUBLISH), ("subscribe", SUBSCRIBE),
                ("await_motion", AWAIT_MOTION), ("await_planes", AWAIT_PLANES), ("release", RELEASE),
                ("await_rows", AWAIT_ROWS),      # NULL here: this Python transport moves whole pictures (the native one has bands)
                ("segment_ownership", _C.c_int)]


class _Plane(_C.Structure):
    _fields_ = [("data", _C.c_void_p), ("stride", _C.c_int32), ("width", _C.c_int32), ("height", _C.c_int32)]


_HOST_GROUP = []


def _host_group():
    """One gloo group per process for host-memory payloads when the default backend is RCCL (created once: every rank must take part)."""
    if not _HOST_GROUP:
        _HOST_GROUP.append(dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else None)
    return _HOST_GROUP[0]


class FrameExchange:
    """Transport behind integration/hip_frames.h.  `lib` = the loaded libohevc_hip.so (ctypes); pass `mode` (an ohhip_frames_mode)
    to the decoder's frames-mode switch (ohhip_set_frames_mode + ohhip_frames_install).  Contexts without a device (record-only,
    the CPU tests' software executor) exchange the HOST planes the decoder registered instead of device pictures."""

    def __init__(self, lib, rank=None, world=None, max_outstanding=32, device=None):
        self.lib = lib
        self.device = device                                       # where device pictures are staged (default: the current GPU)
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.planes_group = None                                   # default group
        # motion fields live in host memory: RCCL cannot carry them
        self.host_group = _host_group() if self.world > 1 else None
        self.pending = {}                                          # index -> (plane works, plane tensors, mvf work, mvf tensor)
        self.outgoing = []                                         # (works, tensors) of published pictures still in flight
        self.max_outstanding = max_outstanding
        self.stats = dict(published=0, subscribed=0, awaited_motion=0, awaited_planes=0, released=0, failed=0, bytes=0)
        self.error = None
        for name, res, args in (("ohevc_ctx_has_device", _C.c_int, [_C.c_void_p]),
                                ("ohevc_pic_planes", _C.c_int, [_C.c_void_p, _C.c_int, _C.c_void_p]),
                                ("ohevc_pic_info", _C.c_int, [_C.c_void_p, _C.c_int] + [_C.c_void_p] * 4),
                                ("ohevc_pic_export", _C.c_int, [_C.c_void_p, _C.c_int, _C.c_int, _C.c_void_p, _C.c_size_t]),
                                ("ohevc_pic_import", _C.c_int, [_C.c_void_p, _C.c_int, _C.c_int, _C.c_void_p, _C.c_size_t]),
                                ("ohevc_tables_host_planes", _C.c_int, [_C.c_void_p, _C.c_int, _C.c_void_p, _C.c_void_p])):
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        self._cb = (_FramesMode.PUBLISH(self._guard(self._publish)), _FramesMode.SUBSCRIBE(self._guard(self._subscribe)),
                    _FramesMode.AWAIT_MOTION(self._guard(self._await_motion)), _FramesMode.AWAIT_PLANES(self._guard(self._await_planes)),
                    _FramesMode.RELEASE(self._guard(self._release)))
        self.mode = _FramesMode(_C.sizeof(_FramesMode), self.rank, self.world, None, *self._cb)

    def _guard(self, fn):
        def call(user, *a):
            try:
                fn(*a)
                return 0
            except Exception as e:                                  # an exception must not unwind through the C decoder
                import traceback
                self.error = e
                traceback.print_exc()
                return -1
        return call

    # ---- staging: the planes of one picture as flat tensors (device: stride x height of the store's layout; host: tight rows)
    def _geometry(self, ctx, slot):
        if self.lib.ohevc_ctx_has_device(ctx):
            pl = (_Plane * 3)()
            if self.lib.ohevc_pic_planes(ctx, slot, pl) != 0:
                raise RuntimeError("ohevc_pic_planes failed")
            return True, [int(p.stride) * int(p.height) for p in pl]
        w, h, cfi, bd = (_C.c_int() for _ in range(4))
        if self.lib.ohevc_pic_info(ctx, slot, _C.byref(w), _C.byref(h), _C.byref(cfi), _C.byref(bd)) != 0:
            raise RuntimeError("ohevc_pic_info failed")
        hs, vs, ps = int(cfi.value in (1, 2)), int(cfi.value == 1), 2 if bd.value > 8 else 1
        self._host_rows = [(h.value, w.value * ps), (h.value >> vs, (w.value >> hs) * ps), (h.value >> vs, (w.value >> hs) * ps)]
        return False, [r * b for r, b in self._host_rows]

    def _alloc(self, on_device, sizes, mvf_bytes):
        dev = torch.device("cpu") if not on_device else self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        # the motion-field message ends with 8 status bytes: [0] != 0 = the owner failed on this picture (hip_frames.h)
        return [torch.empty(n, dtype=torch.uint8, device=dev) for n in sizes], torch.zeros(mvf_bytes + 8, dtype=torch.uint8)

    def _wire(self, planes):
        """What the collective carries: the staging tensors themselves, or host copies when the planes group cannot take device
        tensors (gloo: several ranks on one GPU in the tests; RCCL carries device memory directly)."""
        if self.world > 1 and planes and planes[0].is_cuda and dist.get_backend(self.planes_group) == "gloo":
            return [torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True) for t in planes]
        return planes

    def _host_planes(self, ctx, slot):
        data, ls = (_C.c_void_p * 3)(), (_C.c_int * 3)()
        if self.lib.ohevc_tables_host_planes(ctx, slot, data, ls) != 0:
            raise RuntimeError("ohevc_tables_host_planes failed")
        views = []
        for c, (rows, row_bytes) in enumerate(self._host_rows):
            buf = (_C.c_ubyte * (ls[c] * rows)).from_address(data[c])
            views.append(_np.frombuffer(buf, dtype=_np.uint8).reshape(rows, ls[c])[:, :row_bytes])
        return views

    def _post(self, planes, mvf, src):
        works = [dist.broadcast(t, src=src, group=self.planes_group, async_op=True) for t in planes] if self.world > 1 else []
        mw = dist.broadcast(mvf, src=src, group=self.host_group, async_op=True) if self.world > 1 else None
        self.stats["bytes"] += sum(t.numel() for t in planes) + mvf.numel()
        return works, mw

    # ---- the four callbacks
    def _publish(self, index, ctx, slot, mvf_ptr, mvf_bytes, failed=0):
        on_device, sizes = self._geometry(ctx, slot)
        planes, mvf = self._alloc(on_device, sizes, mvf_bytes)
        if failed:          # the collectives are issued all the same (one per exchanged picture on every rank), marked as failed
            mvf[mvf_bytes] = 1
            for t in planes:
                t.zero_()
            wire = self._wire(planes)
            works, mw = self._post(wire, mvf, self.rank)
            self.outgoing.append((works + ([mw] if mw is not None else []), wire, mvf))
            self.stats["failed"] += 1
            return
        if on_device:
            for c, t in enumerate(planes):
                if self.lib.ohevc_pic_export(ctx, slot, c, t.data_ptr(), t.numel()) != 0:
                    raise RuntimeError("ohevc_pic_export failed")
        else:
            for t, v in zip(planes, self._host_planes(ctx, slot)):
                t.copy_(torch.from_numpy(_np.ascontiguousarray(v)).reshape(-1))
        _C.memmove(mvf.data_ptr(), mvf_ptr, mvf_bytes)
        wire = self._wire(planes)
        if wire is not planes:
            for w, t in zip(wire, planes):
                w.copy_(t)
        works, mw = self._post(wire, mvf, self.rank)
        self.outgoing.append((works + ([mw] if mw is not None else []), wire, mvf))
        while len(self.outgoing) > self.max_outstanding:
            for w in self.outgoing.pop(0)[0]:
                w.wait()
        self.stats["published"] += 1

    def _subscribe(self, index, ctx, slot, mvf_bytes):
        on_device, sizes = self._geometry(ctx, slot)
        planes, mvf = self._alloc(on_device, sizes, mvf_bytes)
        wire = self._wire(planes)
        works, mw = self._post(wire, mvf, index % self.world)
        self.pending[index] = [works, planes, mw, mvf, on_device, wire, False, False]      # ..., planes consumed, motion consumed
        self.stats["subscribed"] += 1

    def _check_failed(self, index, mw, mvf):
        if mw is not None:
            mw.wait()
        if int(mvf[-8]) != 0:
            raise RuntimeError(f"picture {index}: its owner (rank {index % self.world}) reported a decoding failure")

    def _drop_if_consumed(self, index):
        e = self.pending.get(index)
        if e is not None and e[6] and e[7]:                        # planes and motion field both consumed: nothing left to keep
            del self.pending[index]

    def _await_motion(self, index, mvf_ptr, mvf_bytes):
        works, planes, mw, mvf, on_device, wire = self.pending[index][:6]
        self._check_failed(index, mw, mvf)
        _C.memmove(mvf_ptr, mvf.data_ptr(), mvf_bytes)
        self.pending[index][7] = True
        self._drop_if_consumed(index)
        self.stats["awaited_motion"] += 1

    def _release(self, index):
        """The decoder recycled the buffer of remote picture `index`: complete what is in flight for it, drop the staging tensors."""
        e = self.pending.pop(index, None)
        if e is None:
            return
        for w in e[0]:
            w.wait()
        if e[2] is not None:
            e[2].wait()
        self.stats["released"] += 1

    def _await_planes(self, index, ctx, slot):
        works, planes, mw, mvf, on_device, wire = self.pending[index][:6]
        self._check_failed(index, mw, mvf)
        for w in works:
            w.wait()
        if wire is not planes:
            for w, t in zip(wire, planes):
                t.copy_(w)
        if on_device:
            if planes and planes[0].is_cuda:
                torch.cuda.current_stream().synchronize()          # work.wait() only orders torch's stream behind the collective
            for c, t in enumerate(planes):
                if self.lib.ohevc_pic_import(ctx, slot, c, t.data_ptr(), t.numel()) != 0:
                    raise RuntimeError("ohevc_pic_import failed")
        else:
            self._geometry(ctx, slot)
            for t, v in zip(planes, self._host_planes(ctx, slot)):
                v[...] = t.numpy().reshape(v.shape)
        self.pending[index][1] = []                                # the planes are in the store now; the motion field may still be needed
        self.pending[index][5] = []
        self.pending[index][6] = True
        # a picture that is only predicted from (never the collocated picture) keeps its motion field until the decoder recycles the
        # buffer (release); ~1.5 MB per 1080p picture, bounded by the DPB size
        self._drop_if_consumed(index)
        self.stats["awaited_planes"] += 1

    def finish(self):
        """Every collective this rank issued must complete before the process group goes away."""
        for works, *_ in self.outgoing:
            for w in works:
                w.wait()
        self.outgoing.clear()
        for e in self.pending.values():
            for w in e[0]:
                w.wait()
            if e[2] is not None:
                e[2].wait()
        self.pending.clear()
        if self.world > 1:
            dist.barrier()


class NativeFrameTransport:
    """include/ohevc_frames.h: the transport in C inside libohevc_hip.so (ncclBroadcast of planes and motion fields over xGMI, or TCP
    between ranks that share a GPU).  This class only creates / destroys it and hands its callback table to the decoder: no Python runs
    while pictures are exchanged.  `lib` = the loaded product library (ctypes)."""
    WIRE_RCCL, WIRE_SOCKETS = 0, 1

    class Stats(_C.Structure):
        _fields_ = [(n, _C.c_longlong) for n in ("published", "subscribed", "awaited_motion", "awaited_planes", "released", "failed", "bytes", "wire_ranks", "bands_imported")]

    def __init__(self, lib, rank, world, device, wire, rendezvous, timeout_s=120):
        self.lib = lib
        lib.ohevc_frames_transport_create.argtypes = [_C.POINTER(_C.c_void_p), _C.c_int, _C.c_int, _C.c_int, _C.c_int, _C.c_char_p, _C.c_int]
        lib.ohevc_frames_transport_mode.restype = _C.c_void_p
        lib.ohevc_frames_transport_mode.argtypes = [_C.c_void_p]
        lib.ohevc_frames_transport_finish.argtypes = [_C.c_void_p]
        lib.ohevc_frames_transport_destroy.argtypes = [_C.c_void_p]
        lib.ohevc_frames_transport_stats.argtypes = [_C.c_void_p, _C.c_void_p]
        lib.ohevc_last_error.restype = _C.c_char_p
        h = _C.c_void_p()
        if lib.ohevc_frames_transport_create(_C.byref(h), rank, world, device, wire, rendezvous.encode(), timeout_s) != 0:
            raise RuntimeError("native frame transport: " + lib.ohevc_last_error().decode())
        self.h = h
        self.mode = lib.ohevc_frames_transport_mode(h)          # an address: oracle.pystream.Decoder.frames_mode takes it as is
        self.error = None

    @property
    def stats(self):
        st = self.Stats()
        self.lib.ohevc_frames_transport_stats(self.h, _C.byref(st))
        return {n: getattr(st, n) for n, _ in self.Stats._fields_}

    def selftest(self, ctx_handle, src_slot, dst_slot, root, mvf_in):
        """ohevc_frames_transport_selftest: one picture (and `mvf_in`, bytes) from `root` to every rank's dst_slot; returns the motion-field
        bytes that arrived.  Collective: every rank calls it."""
        self.lib.ohevc_frames_transport_selftest.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_int, _C.c_int, _C.c_int, _C.c_char_p, _C.c_void_p, _C.c_size_t]
        out = _C.create_string_buffer(len(mvf_in))
        if self.lib.ohevc_frames_transport_selftest(self.h, ctx_handle, src_slot, dst_slot, root, mvf_in, out, len(mvf_in)) != 0:
            raise RuntimeError("native frame transport self-test: " + self.lib.ohevc_last_error().decode())
        return out.raw

    def set_ownership(self, per_idr_segment):
        """ohhip_frames_mode.segment_ownership: pictures owned per IDR segment (nothing exchanged) instead of per picture"""
        self.lib.ohevc_frames_transport_set_ownership.argtypes = [_C.c_void_p, _C.c_int]
        if self.lib.ohevc_frames_transport_set_ownership(self.h, 1 if per_idr_segment else 0) != 0:
            raise RuntimeError("native frame transport: " + self.lib.ohevc_last_error().decode())

    def finish(self):
        if self.lib.ohevc_frames_transport_finish(self.h) != 0:
            self.error = RuntimeError("native frame transport: " + self.lib.ohevc_last_error().decode())

    def close(self):
        if self.h:
            self.lib.ohevc_frames_transport_destroy(self.h)
            self.h = None
