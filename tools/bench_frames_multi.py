#!/usr/bin/env python3
"""Frame-parallel GOP pipeline over N GPUs (stand-in for BASELINE config 5; synthetic job streams, no bitstream exists here).

    python tools/bench_frames_multi.py --size 1080p --gops 4                      # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_frames_multi.py --size 8k

One process per GPU.  The hierarchical-B stream of openhevc_amd.dist.hierarchical_gop is cut into waves of independent
pictures; each rank reconstructs the pictures it owns through its own ohevc_ctx (planes are torch tensors adopted by the
ctx), then reference pictures are broadcast plane by plane over RCCL (xGMI).  Every picture replays the same synthetic op
list (fresh references each time), so the work per picture is identical and the result is checkable: all ranks must end
with identical DPB contents, and rank 0 compares picture checksums with a single-GPU replay when --check is given."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from openhevc_amd import dist as D, lib as L  # noqa: E402
import stream_exec as X                        # noqa: E402
import synth_stream as S                       # noqa: E402

SIZES = {"416x240": (416, 240, 8), "1080p": (1920, 1080, 8), "4k": (3840, 2160, 10), "8k": (7680, 4320, 10)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1080p", choices=list(SIZES))
    ap.add_argument("--gops", type=int, default=4)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    rank, world = D.init_from_env()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    W, H, bd = SIZES[args.size]
    dt = torch.uint8 if bd == 8 else torch.int16
    ps = 1 if bd == 8 else 2
    ctx = L.Ctx(local)
    rng = np.random.default_rng(4321)
    ops, fops = S.gen_frame_ops(rng, W, H, bd, n_refs=2, intra_frac=0.1)
    pics = D.hierarchical_gop(args.gops, 8)
    slots = {}

    def alloc(idx):
        # 256-byte pitch like ohevc_pic_alloc; the tensor view handed to RCCL covers the whole pitched plane
        planes = []
        for (h, w) in X.chroma_dims(W, H):
            pitch = (w * ps + 255) // 256 * 256 // ps
            planes.append(torch.zeros((h, pitch), dtype=dt, device="cuda")[:, :w])
        slots[idx] = ctx.pic_adopt(planes, W, H, 1, bd)
        return planes

    base_arrays = X.ops_to_arrays(W, H, [0, 1], ops, fops)      # built once; reference slots patched per picture
    base_ref0, base_ref1 = base_arrays["mc"]["ref0"].copy(), base_arrays["mc"]["ref1"].copy()

    def reconstruct(idx, refs, out):
        ref_idx = sorted(refs)
        if not ref_idx:                                   # intra picture: deterministic content, no job stream needed
            g = torch.Generator(device="cuda").manual_seed(99 + idx)
            for t in out:
                t.copy_(torch.randint(0, 1 << bd, t.shape, dtype=dt, device="cuda", generator=g))
            torch.cuda.synchronize()
            return
        rs = np.array([slots[ref_idx[0]], slots[ref_idx[-1]]], dtype=np.int8)
        arrays = dict(base_arrays)
        mc = base_arrays["mc"].copy()
        mc["ref0"], mc["ref1"] = rs[base_ref0], rs[base_ref1]
        arrays["mc"] = mc
        ctx.frame_begin(slots[idx])
        ctx.rec_bulk(**arrays)
        ctx.frame_end()
        ctx.sync()

    runner = D.FrameParallelRunner(alloc=alloc, reconstruct=reconstruct)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    mine = runner.run(pics)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt_s = time.perf_counter() - t0
    sums = {idx: int(sum(int(t.to(torch.int64).sum()) for t in planes)) for idx, planes in mine.items()}
    if world > 1:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, sums)
        sums = {k: v for d in gathered for k, v in d.items()}
    if rank == 0:
        out = {"config": f"{args.size} {bd}-bit hierarchical-B GOP8 x {args.gops}, synthetic job streams, frame-parallel",
               "n_gpus": world, "pictures": len(pics), "seconds": round(dt_s, 4), "pictures_per_s": round(len(pics) / dt_s, 1),
               "Mpixel_per_s": round(len(pics) * W * H / dt_s / 1e6, 1), "broadcast_MB": round(runner.broadcast_bytes / 1e6, 1),
               "waves": len(D.plan_waves(pics)), "checksum": int(sum(sums.values()) % (1 << 61))}
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
