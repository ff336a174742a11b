#!/usr/bin/env python3
"""Why does bench.py's steady-state kernel time differ from the A/B harness?  Replays the bench loop under variations."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from openhevc_amd import lib as L

def run(tag, order, steps, per_step_events, clone=False, variant=-1):
    lib = L.load_library(); lib.ohevc_debug_set_tu_variant(variant)
    n, nblk, per_row = 32, 1 << 20, 512
    g = torch.Generator(device="cuda").manual_seed(1234)
    if order == "plane_first":
        plane = torch.randint(0, 256, (65536, 16384), dtype=torch.uint8, device="cuda", generator=g)
        coeffs = torch.randint(-1024, 1024, (nblk, n, n), dtype=torch.int16, device="cuda", generator=g)
    else:
        coeffs = torch.randint(-1024, 1024, (nblk, n, n), dtype=torch.int16, device="cuda", generator=g)
        plane = torch.randint(0, 256, (65536, 16384), dtype=torch.uint8, device="cuda", generator=g)
    if clone: plane = plane.clone()
    idx = np.arange(nblk); jobs = np.zeros(nblk, L.TU_JOB)
    jobs["x"], jobs["y"], jobs["coeff_off"] = (idx % per_row) * n, (idx // per_row) * n, idx.astype(np.uint32) * n * n
    d_jobs = torch.from_numpy(jobs.view(np.uint8)).cuda()
    planes = L.planes_of([plane, None, None]); st = torch.cuda.current_stream()
    step = lambda: L.dev_tu_batch(planes, 8, 5, L.TU_IDCT, d_jobs.data_ptr(), nblk, coeffs.data_ptr(), st.cuda_stream)
    for _ in range(5): step()
    torch.cuda.synchronize()
    if per_step_events:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for a, b in evs:
            a.record(st); step(); b.record(st)
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
        ts = np.array([a.elapsed_time(b) for a, b in evs])
        print(f"{tag}: per-step mean {ts.mean():.4f} ms (first5 {ts[:5].round(3).tolist()} last5 {ts[-5:].round(3).tolist()}) wall/step {wall/steps*1e3:.4f} ms -> {4294.967296/ts.mean():.0f} GB/s", flush=True)
    else:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(steps): step()
        b.record(st); torch.cuda.synchronize()
        t = a.elapsed_time(b) / steps
        print(f"{tag}: burst mean {t:.4f} ms -> {4294.967296/t:.0f} GB/s", flush=True)
    del plane, coeffs, d_jobs; torch.cuda.empty_cache()

run("bench-like (plane first, per-step events, 20)", "plane_first", 20, True)
run("plane first, burst 20", "plane_first", 20, False)
run("coeffs first, per-step events, 20", "coeffs_first", 20, True)
run("coeffs first, burst 20", "coeffs_first", 20, False)
run("plane first + clone, per-step events 20", "plane_first", 20, True, clone=True)
run("plane first, per-step events, 200", "plane_first", 200, True)
run("plane first, burst 200", "plane_first", 200, False)
run("v0 bench-like", "plane_first", 20, True, variant=0)
run("v144 bench-like", "plane_first", 20, True, variant=144)
