"""Host-side cost of the recording table slots, measurable WITHOUT a GPU: the reference decoder with
(a) empty tables (front-end floor) and (b) the HIP tables in record-only mode (ohevc_debug_set_record_only: jobs are
built and dropped, no pixels).  (b) - (a) = what the slots + recorder cost per picture."""
import os
import sys
import time

os.environ["OHHIP_RECORD_ONLY"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pystream as ps       # noqa: E402


def once(kind, aus):
    """(wall seconds, process CPU seconds) of one pass through a fresh decoder"""
    with ps.Decoder(kind) as d:
        t, c = time.perf_counter(), time.process_time()
        for i, au in enumerate(aus):
            if d.L.ohdec_decode(d.h, au, len(au), i + 1) < 0:
                raise RuntimeError("decode failed")
        return time.perf_counter() - t, time.process_time() - c


def run_pair(aus, repeat=9):
    """null / hip runs INTERLEAVED (this box's speed drifts by 30 % within a minute): the minimum of each, and the median of the paired differences"""
    nulls, hips = [], []
    for _ in range(repeat):
        nulls.append(once("null", aus))
        hips.append(once("hip", aus))
    diffs = sorted(h[1] - n[1] for h, n in zip(hips, nulls))
    return min(n[1] for n in nulls), min(h[1] for h in hips), diffs[len(diffs) // 2]


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "1920x1080"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    bd = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    gop = sys.argv[4] if len(sys.argv) > 4 else "random_access"        # "intra": every picture intra-coded (bench.py's intra_only row)
    w, h = map(int, size.split("x"))
    extra = {}
    if len(sys.argv) > 5 and sys.argv[5] == "natural":                  # encoder-like statistics (bench.NATURAL)
        import bench
        extra = bench.NATURAL
    aus, _ = ps.generate(ps.StreamParams(gop=gop, nframes=n, seed=7, width=w, height=(h + 7) // 8 * 8, log2_ctb=6,
                                         bit_depth=bd, **extra))
    t_null, t_rec, med = run_pair(aus)
    print(f"{size} {bd}-bit {gop} x{n}: front-end only {1e3 * t_null / n:.2f} ms/picture, with recording {1e3 * t_rec / n:.2f} ms/picture, "
          f"recording cost {1e3 * (t_rec - t_null) / n:.2f} ms/picture (CPU time, minima of 9 interleaved runs; median of the paired differences {1e3 * med / n:.2f})")


if __name__ == "__main__":
    main()
