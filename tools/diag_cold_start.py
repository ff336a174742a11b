"""What a freshly opened decoder pays in its first pass at 16 frame threads: the encoder-like 1080p stream of bench.py's decode
block, four passes through one decoder; pictures per second over all passes ("cold") and after the first pass, with the frame buffers page-locked
(default) and left pageable (OHHIP_PIN_FRAMES=0: no hipHostRegister in front of a buffer's first picture, slower copy-backs for ever).

    python tools/diag_cold_start.py [threads] [repeats]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pystream as ps       # noqa: E402

NATURAL = dict(init_qp=32, probs=dict(pred_mode=0.03, skip=0.55, merge_flag=0.7, split_cu=0.3, rqt_root_cbf=0.45, cbf_luma=0.5, cbf_chroma=0.25,
                                      split_transform=0.25, sig_coeff=0.35, last_x=0.5, last_y=0.5))


def timed(kind, aus, th, passes=4, repeat=3):
    best, steady, first = None, 0.0, None
    for _ in range(repeat):
        with ps.Decoder(kind, th, 1) as d:
            t = time.perf_counter()
            n, t_mid, n_mid = 0, None, 0
            for i, au in enumerate(aus * passes):
                if i == len(aus):
                    t_mid, n_mid = time.perf_counter(), n
                r = d.L.ohdec_decode(d.h, au, len(au), i + 1)
                if r < 0:
                    raise RuntimeError(f"decode error {r}")
                n += r
            while True:
                r = d.L.ohdec_flush(d.h)
                if r <= 0:
                    break
                n += r
            t_end = time.perf_counter()
        if best is None or t_end - t < best:
            best, first = t_end - t, t_mid - t
        steady = max(steady, (n - n_mid) / (t_end - t_mid))
    return dict(fps_cold=round(len(aus) * passes / best, 1), fps_after_first_pass=round(steady, 1), first_pass_ms=round(first * 1e3, 2),
                later_pass_ms=round((best - first) / (passes - 1) * 1e3, 2))


def main():
    th = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    aus, _ = ps.generate(ps.StreamParams(gop="random_access", nframes=33, seed=7, width=1920, height=1080, log2_ctb=6, bit_depth=8, **NATURAL))
    out = {"workload": f"1920x1080 encoder-like random-access stream, 33 pictures x 4 passes, {th} frame threads"}
    for label, env in (("page_locked_frame_buffers", None), ("pageable_frame_buffers", "0"), ("page_locked_again", None)):
        if env is None:
            os.environ.pop("OHHIP_PIN_FRAMES", None)
        else:
            os.environ["OHHIP_PIN_FRAMES"] = env
        out[label] = timed("hip", aus, th, repeat=rep)
    if ps.have("sse"):
        out["reference_sse"] = timed("sse", aus, th, repeat=rep)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
