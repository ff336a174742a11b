"""Host-side cost of the recording table slots, measurable WITHOUT a GPU: the reference decoder with
(a) empty tables (front-end floor) and (b) the HIP tables in record-only mode (ohevc_debug_set_record_only: jobs are
built and dropped, no pixels).  (b) - (a) = what the slots + recorder cost per picture."""
import os
import sys
import time

os.environ["OHHIP_RECORD_ONLY"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pystream as ps       # noqa: E402


def run(kind, aus, repeat=3):
    best = 1e9
    for _ in range(repeat):
        with ps.Decoder(kind) as d:
            t = time.perf_counter()
            for i, au in enumerate(aus):
                if d.L.ohdec_decode(d.h, au, len(au), i + 1) < 0:
                    raise RuntimeError("decode failed")
            best = min(best, time.perf_counter() - t)
    return best


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "1920x1080"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    bd = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    w, h = map(int, size.split("x"))
    aus, _ = ps.generate(ps.StreamParams(gop="random_access", nframes=n, seed=7, width=w, height=(h + 7) // 8 * 8, log2_ctb=6,
                                         bit_depth=bd))
    t_null = run("null", aus)
    t_rec = run("hip", aus)
    print(f"{size} {bd}-bit x{n}: front-end only {1e3 * t_null / n:.2f} ms/picture, with recording {1e3 * t_rec / n:.2f} ms/picture, "
          f"recording cost {1e3 * (t_rec - t_null) / n:.2f} ms/picture")


if __name__ == "__main__":
    main()
