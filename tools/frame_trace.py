#!/usr/bin/env python3
"""Host-side timeline of every picture of a decode (integration/hip_backend.h: ohhip_options.trace_path / OHHIP_TRACE_FRAMES).

    OHHIP_TRACE_FRAMES=/tmp/ft.txt python tools/diag_overlap.py decode 16 natural
    python tools/frame_trace.py /tmp/ft.txt [--dump]

Each line of the trace: back-end id, thread, POC, t_start (hevc_frame_start: the thread begins to parse the picture's slice data),
t_hook (the frame-end hook begins: parsing is over), t_issued (stage + upload + launches issued, incl. the wait for the reference pictures'
frame ends), t_end (device work and copy-back have landed).  Prints, for the longest-lived back end: pictures, wall time, fps, the mean of the
three phases, and how the decoding threads' time divides between parsing, issuing / waiting for references, waiting for the device, and
idling between pictures (waiting for the application thread to hand out the next access unit)."""
import collections
import json
import sys


def load(path):
    by = collections.defaultdict(list)
    for line in open(path):
        f = line.split()
        if len(f) == 7:
            by[int(f[0])].append((int(f[1]), int(f[2])) + tuple(float(x) for x in f[3:]))
    return by


def summarise(recs, tail=0.75):
    recs = sorted(recs, key=lambda r: r[2])
    t0, t1 = recs[0][2], max(r[5] for r in recs)
    lo = t0 + (t1 - t0) * (1 - tail)                   # the start-up (contexts, streams, page locks, first pictures) is left out
    st = [r for r in recs if r[2] >= lo]
    n = len(st)
    span = max(r[5] for r in st) - min(r[2] for r in st)
    parse = sum(r[3] - r[2] for r in st) / n
    issue = sum(r[4] - r[3] for r in st) / n
    wait = sum(r[5] - r[4] for r in st) / n
    threads = len({r[0] for r in st})
    busy = (parse + issue + wait) * n
    # how many pictures are inside their hook at once, time-weighted
    ev = sorted([(r[3], 1) for r in st] + [(r[5], -1) for r in st])
    inside, last, acc = 0, ev[0][0], collections.Counter()
    for t, d in ev:
        acc[inside] += t - last
        inside += d
        last = t
    tot = sum(acc.values())
    return dict(pictures=n, threads=threads, span_ms=round(span * 1e3, 2), fps=round(n / span, 1), parse_ms=round(parse * 1e3, 3),
                issue_and_reference_wait_ms=round(issue * 1e3, 3), device_and_copy_back_wait_ms=round(wait * 1e3, 3),
                idle_between_pictures_ms=round((span * threads - busy) / n * 1e3, 3),
                share_of_time_with_n_hooks_open={k: round(v / tot, 3) for k, v in sorted(acc.items()) if v / tot >= 0.005})


if __name__ == "__main__":
    by = load(sys.argv[1])
    be = max(by, key=lambda k: len(by[k]))
    print(json.dumps(summarise(by[be])))
    if "--dump" in sys.argv:
        t0 = min(r[2] for r in by[be])
        for r in sorted(by[be], key=lambda r: r[3]):
            print(f"tid {r[0]:5d} poc {r[1]:4d}  start {1e3 * (r[2] - t0):9.3f}  parse {1e3 * (r[3] - r[2]):7.3f}  issue {1e3 * (r[4] - r[3]):7.3f}  wait {1e3 * (r[5] - r[4]):7.3f}  end {1e3 * (r[5] - t0):9.3f}")
