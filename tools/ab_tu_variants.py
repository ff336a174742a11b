#!/usr/bin/env python3
"""Within-process interleaved A/B of the IDCT+add kernel variants (ohevc_debug_set_tu_variant).
Checks that every variant produces bit-identical planes, then times them in interleaved rounds (median / min)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openhevc_amd import lib as L  # noqa: E402


def setup(log2, bd, nblk):
    n = 1 << log2
    per_row = 16384 // n
    H = nblk // per_row * n
    g = torch.Generator(device="cuda").manual_seed(1234)
    plane = torch.randint(0, 1 << bd, (H, 16384), dtype=torch.uint8 if bd == 8 else torch.int16, device="cuda", generator=g)
    coeffs = torch.randint(-1024, 1024, (nblk, n, n), dtype=torch.int16, device="cuda", generator=g)
    idx = np.arange(nblk)
    jobs = np.zeros(nblk, L.TU_JOB)
    jobs["x"], jobs["y"], jobs["coeff_off"] = (idx % per_row) * n, (idx // per_row) * n, idx.astype(np.uint32) * n * n
    d_jobs = torch.from_numpy(jobs.view(np.uint8)).cuda()
    return plane, coeffs, d_jobs


def main():
    lib = L.load_library()
    variants = sys.argv[1].split(",") if len(sys.argv) > 1 else "0,4,5,12,4@1024,4@4096,12@1280".split(",")

    def select(v):                       # "variant" or "variant@pipe_workgroups"
        var, _, wgs = v.partition("@")
        lib.ohevc_debug_set_tu_pipe_workgroups(int(wgs) if wgs else 2048)
        lib.ohevc_debug_set_tu_variant(int(var))

    rounds = 8
    results = {}
    configs = [(5, 8, 1 << 20), (5, 10, 1 << 20), (4, 8, 1 << 22), (3, 8, 1 << 24)]
    if len(sys.argv) > 2 and sys.argv[2] == "32":          # only the 32x32 configurations
        configs = configs[:2]
    for (log2, bd, nblk) in configs:
        n = 1 << log2
        plane0, coeffs, d_jobs = setup(log2, bd, nblk)
        st = torch.cuda.current_stream()
        outs = {}
        for v in variants:
            select(v)
            p = plane0.clone()
            L.dev_tu_batch(L.planes_of([p, None, None]), bd, log2, L.TU_IDCT, d_jobs.data_ptr(), nblk, coeffs.data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
            outs[v] = p
        same = all(torch.equal(outs[variants[0]], outs[v]) for v in variants)
        del outs
        work = plane0.clone()
        planes = L.planes_of([work, None, None])
        times = {v: [] for v in variants}
        BURST = 12          # launches back to back per sample: steady-state clocks (isolated launches run ~10% faster)
        for r in range(rounds + 2):
            for v in variants:
                select(v)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                for _ in range(BURST):
                    L.dev_tu_batch(planes, bd, log2, L.TU_IDCT, d_jobs.data_ptr(), nblk, coeffs.data_ptr(), st.cuda_stream)
                b.record(st)
                torch.cuda.synchronize()
                if r >= 2:
                    times[v].append(a.elapsed_time(b) / BURST)
        bytes_ = nblk * n * n * (2 + 2 * (2 if bd > 8 else 1))
        row = {}
        for v in variants:
            t = np.array(times[v])
            row[f"v{v}"] = {"median_ms": round(float(np.median(t)), 4), "min_ms": round(float(t.min()), 4),
                            "GBps_median": round(bytes_ / np.median(t) / 1e6, 1)}
        results[f"{n}x{n}_{bd}bit"] = {"identical": same, **row}
        print(f"{n}x{n} {bd}-bit identical={same} " + " ".join(f"v{v}:{row[f'v{v}']['GBps_median']}GB/s" for v in variants), flush=True)
        del plane0, coeffs, d_jobs, work
        torch.cuda.empty_cache()
    select("-1")
    print(json.dumps(results))


if __name__ == "__main__":
    main()
