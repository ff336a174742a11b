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


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (set by torch.distributed.run). Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend, **kw)
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
