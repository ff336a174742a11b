#!/usr/bin/env python3
"""Frame-pipeline bench on synthetic job streams (stand-in for BASELINE configs 3/4: no HEVC bitstream exists here).

One picture's worth of legal reconstruction work (tools/synth_stream.py: RA-like mix of bi/uni inter CUs, intra CUs,
residual TUs, deblocking edges, SAO) is recorded into the ctx layer and executed:  MC -> inter residuals -> intra levels
-> vertical edges -> horizontal edges -> SAO.  Reports decoded Mpixel/s (luma W*H per picture) for (a) device execution
incl. the job/coefficient upload, with records prepared in host memory, and (b) the reference's own C tables executing the
same op list on one host thread (through oracle/_ref, ctypes dispatch included -- a floor for the CPU, not a tuned run).
Writes one JSON line per configuration.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from openhevc_amd import lib as L  # noqa: E402
import stream_exec as X            # noqa: E402
import synth_stream as S           # noqa: E402


def run(name, W, H, bd, frames, check, intra_frac=0.1, cpu=True):
    rng = np.random.default_rng(1234)
    dt = np.uint16 if bd > 8 else np.uint8
    dims = X.chroma_dims(W, H)
    refs = [[rng.integers(0, 1 << bd, size=d).astype(dt) for d in dims] for _ in range(2)]
    cur0 = [rng.integers(0, 1 << bd, size=d).astype(dt) for d in dims]
    t0 = time.perf_counter()
    ops, fops = S.gen_frame_ops(rng, W, H, bd, n_refs=2, intra_frac=intra_frac)
    t_gen = time.perf_counter() - t0
    ctx = L.Ctx(0)
    slots = []
    for r in refs:
        s = ctx.pic_alloc(W, H, 1, bd); ctx.pic_upload(s, r); slots.append(s)
    cur = ctx.pic_alloc(W, H, 1, bd)
    arrays = X.ops_to_arrays(W, H, slots, ops, fops)
    t_rec, t_exec = [], []
    for f in range(frames + 2):
        ctx.pic_upload(cur, cur0)
        ctx.sync()
        a = time.perf_counter()
        ctx.frame_begin(cur)
        ctx.rec_bulk(**arrays)
        b = time.perf_counter()
        ctx.frame_end()
        ctx.sync()
        c = time.perf_counter()
        if f >= 2:
            t_rec.append(b - a); t_exec.append(c - b)
    st = ctx.stats()
    out = {"config": name, "width": W, "height": H, "bit_depth": bd, "ops": {k: int(len(v)) for k, v in arrays.items() if k != "tu_coeffs"},
           "coeff_bytes": int(arrays["tu_coeffs"].nbytes), "stats": st,
           "record_ms": round(1e3 * float(np.median(t_rec)), 3), "execute_ms": round(1e3 * float(np.median(t_exec)), 3),
           "gpu_Mpixel_per_s_execute": round(W * H / float(np.median(t_exec)) / 1e6, 1),
           "gpu_fps_execute": round(1.0 / float(np.median(t_exec)), 1)}
    if check or cpu:
        from oracle import pyoracle as po
        lib = po.load("ref") or po.load("oracle")
        t0 = time.perf_counter()
        want = X.run_oracle(lib, po, bd, W, H, [p.copy() for p in cur0], refs, ops, fops)
        t_cpu = time.perf_counter() - t0
        out["cpu_reference_1thread_Mpixel_per_s"] = round(W * H / t_cpu / 1e6, 2)
        out["cpu_reference_kind"] = "reference" if po.load("ref") else "port"
        if check:
            got = ctx.pic_download(cur, dims, dt)
            out["bit_exact_vs_oracle"] = bool(all(np.array_equal(g, w) for g, w in zip(got, want)))
    ctx.close()
    out["generate_s"] = round(t_gen, 2)
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "small"):
        run("416x240 8-bit synthetic RA-like job stream", 416, 240, 8, 10, True)
    if which in ("all", "1080p"):
        run("1080p 8-bit synthetic RA-like job stream (config 3 stand-in)", 1920, 1080, 8, 10, True)
    if which in ("all", "4k"):
        run("4K Main10 synthetic RA-like job stream (config 4 stand-in)", 3840, 2160, 10, 5, True)
    if which in ("8k",):
        run("8K Main10 synthetic RA-like job stream (config 5 stand-in, 1 GPU)", 7680, 4320, 10, 3, False, cpu=False)
