"""-m gpu: the frame-parallel decoder over processes on a real device.  The GPU box has one GPU, so both ranks use cuda:0 and the planes
travel through a gloo group (host-staged, openhevc_amd.dist.FrameExchange._wire); everything else - slice data parsed by the owner only,
ohevc_pic_export / ohevc_pic_import on device pictures, the waits for motion fields and planes - is the path a multi-GPU run takes
with RCCL.  Every picture must equal the single-process decoder's (the committed digests)."""
import os

import pytest
import torch.multiprocessing as mp

from test_dist_cpu import decoder_worker, free_port

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
