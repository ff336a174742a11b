#!/usr/bin/env python3
"""Frame-parallel scaling model of the decoder: what N ranks make of a stream, from numbers measured on ONE GPU.

    python tools/scaling_model.py measure [WxH] [bit_depth] [pictures]   -> JSON: per-picture ms of the stream bench.py's `frames` object decodes
    python tools/scaling_model.py predict <measure.json> [GB/s per exchange]  -> JSON: fps at 1, 2, 4, 8 ranks

The protocol (integration/hip_frames.h): picture k (decoding order) is owned by rank k mod N; the owner parses its slice data and reconstructs
it, everybody else only its headers.  A rank takes its pictures in decoding order.  A picture can start when its rank is free AND the motion
field of every picture it predicts from has arrived (ff_hevc_frame_rps waits for them before the parse starts, as the reference's frame
threads wait for the collocated picture); it can be launched when the planes of those pictures have arrived, as far down as its motion
vectors reach.  The owner publishes at its frame end: motion field first, then the bands of CTU rows.
Model: list scheduling over that dependency graph with
    t_own[k]    ms a rank spends on a picture it owns (measured: the hooked decoder, one thread, steady state)
    t_skip[k]   ms a rank spends on a picture it does not own (headers only; measured: the front end, slice data skipped ~ 0.05 ms)
    x_mvf, x_pl ms until the motion field / the planes of a published picture are usable on another rank = bytes / bandwidth
Nothing else: no contention between ranks for host cores (one thread each), none for the wire (a picture's 150 MB at 8K are ~3 ms of a
48 GB/s broadcast against ~22 ms of parsing)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stream(size, bd, pictures):
    import bench
    from oracle import pystream as ps
    w, h = size
    p = ps.StreamParams(gop="random_access", nframes=pictures, seed=4242, width=w, height=(h + 7) // 8 * 8, bit_depth=bd, log2_ctb=6, nonref_leaves=1, **bench.NATURAL)
    aus, _ = ps.generate(p)
    pics = ps.plan_gop(p)
    poc_to_idx = {pic.poc: i for i, pic in enumerate(pics)}
    refs = [[poc_to_idx[q] for q, used in list(pic.rps_neg) + list(pic.rps_pos) if used and q in poc_to_idx] for pic in pics]
    exchanged = [pic.nal_type != ps.NAL_TRAIL_N for pic in pics]
    return aus, refs, exchanged


def measure(size, bd, pictures):
    from oracle import pystream as ps
    aus, refs, exchanged = stream(size, bd, pictures)
    out = {"size": list(size), "bit_depth": bd, "pictures": pictures, "bytes_per_picture": sum(map(len, aus)) // len(aus), "refs": refs, "exchanged": exchanged}
    for kind in ("null", "hip"):
        if not ps.have(kind):
            continue
        if kind == "null":
            ps._load("null").ohnull_set_await(0)
        best = None
        for _ in range(3):
            with ps.Decoder(kind, 1, 1) as d:
                for i, au in enumerate(aus):                       # first pass: start-up
                    d.L.ohdec_decode(d.h, au, len(au), i + 1)
                ms = []
                for i, au in enumerate(aus):                       # second pass through the same decoder: steady state, per access unit
                    t = time.perf_counter()
                    d.L.ohdec_decode(d.h, au, len(au), 100 + i)
                    ms.append(1e3 * (time.perf_counter() - t))
                while d.L.ohdec_flush(d.h) > 0:
                    pass
            if best is None or sum(ms) < sum(best):
                best = ms
        out["ms_" + kind] = [round(v, 3) for v in best]
    w, h = size
    ps_ = 2 if bd > 8 else 1
    stride = lambda n: (n * ps_ + 255) & ~255
    out["plane_bytes"] = stride(w) * ((h + 7) // 8 * 8) + 2 * stride(w // 2) * (((h + 7) // 8 * 8) // 2)
    out["mvf_bytes"] = ((w + 3) // 4) * (((h + 7) // 8 * 8 + 3) // 4) * 24
    print(json.dumps(out))


def simulate(m, n_ranks, gbs, passes=8, skip_ms=0.05):
    """fps of `passes` repetitions of the measured stream (each starts with an IDR picture: no dependency across repetitions, but a rank takes
    its pictures in order) on n_ranks ranks"""
    t_own = m["ms_hip"]
    refs, exch = m["refs"], m["exchanged"]
    x_mvf = m["mvf_bytes"] / (gbs * 1e6) if n_ranks > 1 else 0.0        # ms
    x_pl = (m["mvf_bytes"] + m["plane_bytes"]) / (gbs * 1e6) if n_ranks > 1 else 0.0
    n = len(t_own)
    free = [0.0] * n_ranks
    finish = {}
    for p in range(passes):
        for k in range(n):
            g = p * n + k
            owner = g % n_ranks
            for r in range(n_ranks):
                if r != owner:
                    free[r] += skip_ms
            start = free[owner]
            launch_ready = 0.0
            for q in refs[k]:
                gq = p * n + q
                remote = gq % n_ranks != owner
                start = max(start, finish[gq] + (x_mvf if remote else 0.0))
                launch_ready = max(launch_ready, finish[gq] + (x_pl if remote else 0.0))
            end = max(start + t_own[k], launch_ready)
            # the owner exports + posts its picture (host-synchronous d2d copies per band; the wire itself is asynchronous)
            end += 0.0 if n_ranks == 1 or not exch[k] else (m["plane_bytes"] / 2.0e9)      # ~2 TB/s device copies: 0.05 ms at 8K
            finish[g] = end
            free[owner] = end
    total = max(free)
    return passes * n / (total / 1e3)


def predict(path, gbs):
    m = json.load(open(path))
    out = {"model": "list scheduling over the decoding-order dependency graph (tools/scaling_model.py)", "exchange_GB_per_s": gbs,
           "stream": f"{m['size'][0]}x{m['size'][1]} {m['bit_depth']}-bit, {m['pictures']} pictures, {m['bytes_per_picture']} bytes/picture",
           "ms_per_picture_owned_mean": round(sum(m["ms_hip"]) / len(m["ms_hip"]), 2),
           "ms_per_picture_front_end_alone_mean": round(sum(m["ms_null"]) / len(m["ms_null"]), 2) if "ms_null" in m else None,
           "bytes_per_exchanged_picture": m["plane_bytes"] + m["mvf_bytes"], "fps": {}}
    for n in (1, 2, 4, 8):
        out["fps"][str(n)] = round(simulate(m, n, gbs), 2)
    out["speedup"] = {k: round(v / out["fps"]["1"], 2) for k, v in out["fps"].items()}
    print(json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "measure":
        size = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "7680x4320").split("x"))
        measure(size, int(sys.argv[3]) if len(sys.argv) > 3 else 10, int(sys.argv[4]) if len(sys.argv) > 4 else 9)
    else:
        predict(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 48.0)
