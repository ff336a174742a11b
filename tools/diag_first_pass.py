"""Where a freshly opened decoder's FIRST pass goes at 16 frame threads: the per-picture host timeline (OHHIP_TRACE_FRAMES) of the encoder-like 1080p
stream, two passes through one decoder; per pass: wall time and the mean / max of a picture's parse, issue (incl. waits for reference pictures'
frame ends) and device-wait phases, and the ten slowest hooks of the first pass with what they were.

    python tools/diag_first_pass.py [threads]
"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pystream as ps       # noqa: E402
from diag_cold_start import NATURAL     # noqa: E402


def main():
    th = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    aus, _ = ps.generate(ps.StreamParams(gop="random_access", nframes=33, seed=7, width=1920, height=1080, log2_ctb=6, bit_depth=8, **NATURAL))
    with ps.Decoder("hip", th, 1) as d:          # (the process's first decoder pays the runtime's own start-up: not the one that is traced)
        for i, au in enumerate(aus):
            d.L.ohdec_decode(d.h, au, len(au), i + 1)
        while d.L.ohdec_flush(d.h) > 0:
            pass
    path = tempfile.mktemp(suffix=".txt")
    os.environ["OHHIP_TRACE_FRAMES"] = path
    marks = []
    with ps.Decoder("hip", th, 1) as d:
        t0 = time.clock_gettime(time.CLOCK_MONOTONIC)
        for p in range(2):
            marks.append(time.clock_gettime(time.CLOCK_MONOTONIC))
            for i, au in enumerate(aus):
                d.L.ohdec_decode(d.h, au, len(au), p * len(aus) + i + 1)
        while d.L.ohdec_flush(d.h) > 0:
            pass
        t_end = time.clock_gettime(time.CLOCK_MONOTONIC)
    os.environ.pop("OHHIP_TRACE_FRAMES")
    recs = []
    for line in open(path):
        f = line.split()
        if len(f) == 7:
            recs.append((int(f[1]), int(f[2])) + tuple(float(x) for x in f[3:]))
    recs.sort(key=lambda r: r[2])
    out = {"workload": f"1920x1080 encoder-like stream, 33 pictures x 2 passes, {th} frame threads, second decoder of the process",
           "wall_ms": round((t_end - t0) * 1e3, 2), "pictures_traced": len(recs)}
    half = len(recs) // 2
    for name, part in (("first_pass", recs[:half]), ("second_pass", recs[half:])):
        if not part:
            continue
        ph = lambda a, b: [1e3 * (r[b] - r[a]) for r in part]      # noqa: E731
        parse, issue, wait = ph(2, 3), ph(3, 4), ph(4, 5)
        out[name] = {"span_ms": round(1e3 * (max(r[5] for r in part) - min(r[2] for r in part)), 2),
                     "parse_ms_mean_max": [round(sum(parse) / len(parse), 3), round(max(parse), 3)],
                     "issue_ms_mean_max": [round(sum(issue) / len(issue), 3), round(max(issue), 3)],
                     "device_wait_ms_mean_max": [round(sum(wait) / len(wait), 3), round(max(wait), 3)],
                     "threads": len({r[0] for r in part})}
    first = recs[:half]
    t_first = min(r[2] for r in first) if first else 0
    out["first_pass_slowest_hooks"] = [dict(poc=r[1], thread=r[0], parse_start_ms=round(1e3 * (r[2] - t_first), 2), parse_ms=round(1e3 * (r[3] - r[2]), 2),
                                            issue_ms=round(1e3 * (r[4] - r[3]), 2), device_wait_ms=round(1e3 * (r[5] - r[4]), 2))
                                       for r in sorted(first, key=lambda r: r[3] - r[5])[:10]]
    if os.environ.get("DIAG_DUMP"):      # every picture of the first pass: [poc, thread, parse start, parse end, issued, landed] in ms from the first parse start
        out["first_pass_pictures"] = [[r[1], r[0]] + [round(1e3 * (r[k] - t_first), 2) for k in (2, 3, 4, 5)] for r in first]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
