#!/usr/bin/env python3
"""Do the kernels of different decoding threads overlap on the device?

    rocprofv3 --kernel-trace -d OUT -o t -- python tools/diag_overlap.py decode <threads> [natural]
    python tools/diag_overlap.py analyze OUT/t_results.db

decode : synthetic 1080p random-access stream (33 pictures) through the hooked reference decoder with <threads> frame threads; prints fps.
analyze: from the kernel-dispatch timestamps: kernels, sum of their durations, the union of their busy intervals (= device time with at
         least one kernel running), the span from the first start to the last end, and how many hardware queues carried them."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def decode(threads, natural):
    from oracle import pystream as ps
    natural = natural or os.environ.get("DIAG_NATURAL") == "1"
    kw = dict(gop=os.environ.get("DIAG_GOP", "random_access"), nframes=33, seed=7, width=1920, height=1080, log2_ctb=6)      # DIAG_GOP=intra: every picture intra
    if kw["gop"] == "intra":
        kw["nframes"] = 17
    if natural:
        kw.update(init_qp=32, probs=dict(pred_mode=0.03, skip=0.55, merge_flag=0.7, split_cu=0.3, rqt_root_cbf=0.45, cbf_luma=0.5, cbf_chroma=0.25,
                                          split_transform=0.25, sig_coeff=0.35, last_x=0.5, last_y=0.5))
    if os.environ.get("DIAG_DENSE") == "1":                             # bench.py's qp22-like density instead
        kw.update(ps.DENSE_QP22)
    aus, _ = ps.generate(ps.StreamParams(**kw))
    kind = os.environ.get("DIAG_KIND", "hip")                        # "hipemu" with OHHIP_RECORD_ONLY=1: the hooks' recording alone, without a device (CPU box)
    for a in sys.argv[3:]:                                           # name=value: ohevc_debug_set_<name>(value) of the product library (A/B runs)
        if "=" in a:
            getattr(ps._product_lib(), "ohevc_debug_set_" + a.split("=")[0])(int(a.split("=")[1]))
    ps.decode_stream(kind, aus[:9], threads, 1)                      # warm-up: library load, allocations
    passes = int(os.environ.get("DIAG_PASSES", "4"))                 # the stream several times through ONE decoder: steady state
    best = None
    for _ in range(2):
        # (no picture is copied out into Python: ps.decode_stream's per-picture numpy copies cap the main thread at ~1500 pictures a second
        # and were what the first overlap profiles of round 3 measured)
        with ps.Decoder(kind, threads, 1) as d:
            t = time.perf_counter()
            n = 0
            for i, au in enumerate(aus * passes):
                r = d.L.ohdec_decode(d.h, au, len(au), i + 1)
                assert r >= 0, r
                n += r
            while True:
                r = d.L.ohdec_flush(d.h)
                if r <= 0:
                    break
                n += r
            dt = time.perf_counter() - t
            import ctypes
            tm = (ctypes.c_double * 3)()
            if hasattr(d.L, "ohdec_times"):
                d.L.ohdec_times(ctypes.c_void_p(d.h) if isinstance(d.h, int) else d.h, tm)
                print(json.dumps(dict(application_thread=dict(wall_s=round(dt, 4), in_decoder_call_s=round(tm[0], 4), in_fetch_output_s=round(tm[1], 4), calls=int(tm[2])))))
        out = [None] * n
        best = dt if best is None else min(best, dt)
    print(json.dumps(dict(threads=threads, natural=bool(natural), passes=passes, pictures=len(out), fps=round(len(out) / best, 1))))


def analyze(db):
    import sqlite3
    con = sqlite3.connect(db)
    rows = con.execute("select start, end, queue_id, stream_id, name from kernels order by start").fetchall()
    rows = [r for r in rows if "ohevc" in r[4]]
    if not rows:
        print("no ohevc kernels in the trace")
        return
    total = sum(e - s for s, e, *_ in rows)
    union, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    for s, e, *_ in rows[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    span = max(r[1] for r in rows) - rows[0][0]
    gaps = sorted(rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1))
    # the long kernels (the intra chains: one workgroup each, milliseconds): how many of them run at the same time, and on which queues
    longk = [r for r in rows if r[1] - r[0] > 200000]
    if longk:
        ev = sorted([(r[0], 1) for r in longk] + [(r[1], -1) for r in longk])
        cur, last, at = 0, ev[0][0], {}
        for t, d in ev:
            at[cur] = at.get(cur, 0) + (t - last)
            cur += d; last = t
        busy = sum(v for k, v in at.items() if k > 0)
        per_queue = {}
        for r in longk:
            per_queue[r[2]] = per_queue.get(r[2], 0) + 1
        print(json.dumps(dict(long_kernels=len(longk), mean_long_ms=round(sum(r[1] - r[0] for r in longk) / len(longk) / 1e6, 3),
                              share_of_time_with_n_long_kernels_running={k: round(v / busy, 3) for k, v in sorted(at.items()) if k > 0},
                              long_kernels_per_queue=per_queue, names=sorted({r[4].split("ohevc::")[-1][:40] for r in longk}))))
    print(json.dumps(dict(kernels=len(rows), queues=len({r[2] for r in rows}), streams=len({r[3] for r in rows}),
                          sum_kernel_ms=round(total / 1e6, 3), union_busy_ms=round(union / 1e6, 3), span_ms=round(span / 1e6, 3),
                          mean_kernel_us=round(total / len(rows) / 1e3, 2), overlap_factor=round(total / union, 3),
                          busy_share_of_span=round(union / span, 3))))


def chain(db):
    """Per kernel: launches, mean duration, and the mean idle time of its stream until the next kernel on that stream starts - what one
    step of a dependent chain (the per-level launches of the intra blocks) costs beyond the kernel itself."""
    import re
    import sqlite3
    con = sqlite3.connect(db)
    rows = con.execute("select start, end, queue_id, stream_id, name from kernels order by start").fetchall()
    by_stream = {}
    for r in rows:
        by_stream.setdefault(r[3], []).append(r)
    acc = {}
    for st, rs in by_stream.items():
        for i, (s, e, q, _, name) in enumerate(rs):
            if "ohevc" not in name:
                continue
            short = re.sub(r"<.*", "", name.split("ohevc::")[-1])
            a = acc.setdefault(short, [0, 0, 0, 0])
            a[0] += 1; a[1] += e - s
            if i + 1 < len(rs) and "ohevc" in rs[i + 1][4] and rs[i + 1][0] - e < 200000:       # gaps above 0.2 ms: the host was not ready
                a[2] += 1; a[3] += max(rs[i + 1][0] - e, 0)
    for k, (n, dur, ng, gap) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(json.dumps(dict(kernel=k, launches=n, mean_us=round(dur / n / 1e3, 2), total_ms=round(dur / 1e6, 3),
                              mean_gap_to_next_on_stream_us=round(gap / ng / 1e3, 2) if ng else None)))


def dump(db, out):
    """The whole trace as a compact gzip'd CSV: kernel dispatches (k), memory copies (m) and - when the HIP runtime was traced - API calls (a):
    start, end, queue / thread, stream, kind, name."""
    import gzip
    import re
    import sqlite3
    con = sqlite3.connect(db)
    names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    info = {}
    with gzip.open(out, "wt") as f:
        for s, e, q, st, name in con.execute("select start, end, queue_id, stream_id, name from kernels order by start"):
            short = re.sub(r"<.*", "", name.split("ohevc::")[-1]).split("(")[0][-40:]
            f.write(f"{s},{e},{q},{st},k,{short}\n")
        for v, kind in (("memory_copies", "m"), ("regions", "a")):
            if v not in names:
                continue
            cols = [c[1] for c in con.execute(f"pragma table_info({v})")]
            info[v] = cols
            want = [c for c in ("start", "end", "tid", "queue_id", "stream_id", "name", "size") if c in cols]
            try:
                for row in con.execute(f"select {','.join(want)} from {v} order by start"):
                    d = dict(zip(want, row))
                    f.write(f"{d.get('start')},{d.get('end')},{d.get('tid', d.get('queue_id'))},{d.get('stream_id')},{kind},{d.get('name')}:{d.get('size', '')}\n")
            except Exception as ex:      # a view whose base tables are missing in this trace
                info[v + "_error"] = str(ex)
    print(json.dumps(dict(tables=[n for n in names if not n.startswith("rocpd_")][:40], columns=info)))


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "chain":
        chain(sys.argv[2])
    elif sys.argv[1] == "decode":
        decode(int(sys.argv[2]), "natural" in sys.argv[3:])
    else:
        analyze(sys.argv[2])
