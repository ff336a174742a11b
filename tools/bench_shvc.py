"""Two-layer (SHVC, spatial x2) decode: both layers of a synthetic stream through the reference's pair of decoders (gpac/modules/openhevc_dec/
openHevcWrapper.c:47-156) - untouched C, the reference as shipped on x86 (SSE4), and the same front end with the gfx950 back end (the
enhancement layer's back end shares the base layer's device picture store; the inter-layer picture is resampled on the device).

    python tools/bench_shvc.py [--size 1920x1088] [--frames 17] [--passes 3] [--kinds c,sse,hip]

--size is the ENHANCEMENT layer; the base layer is half of it in each direction.  One decoding thread per layer (the reference's frame-threaded
two-decoder mode gives run-to-run different pictures on its own tables).  Prints one JSON line: access units
per second (= pictures per second of each layer), wall clock, entropy decoding of both layers and the copy-back of every picture included."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                      # noqa: E402
from oracle import pystream as ps       # noqa: E402

NATURAL = dict(init_qp=32, probs=dict(pred_mode=0.03, skip=0.55, merge_flag=0.7, split_cu=0.3, rqt_root_cbf=0.45, cbf_luma=0.5, cbf_chroma=0.25,
                                      split_transform=0.25, sig_coeff=0.35, last_x=0.5, last_y=0.5))


def timed(kind, aus, passes, repeat=2):
    best, out = None, None
    for _ in range(repeat):
        bl = ps.Decoder(kind)
        el = ps.Decoder(kind, decoder_id=1, base=bl)
        bl.set_active_layer(1)
        try:
            t = time.perf_counter()
            n = [0, 0]
            for i, au in enumerate(aus * passes):
                for k, d in enumerate((bl, el)):
                    if k:
                        el.take_base(bl)
                    r = d.L.ohdec_decode(d.h, au, len(au), i + 1)
                    if r < 0:
                        raise RuntimeError(f"decode error {r} ({kind}, layer {k})")
                    n[k] += r
            for k, d in enumerate((bl, el)):
                while True:
                    r = d.L.ohdec_flush(d.h)
                    if r <= 0:
                        break
                    n[k] += r
            dt = time.perf_counter() - t
        finally:
            el.close()
            bl.close()
        best = dt if best is None else min(best, dt)
        out = n
    return best, out


def run(size=(1920, 1088), frames=17, passes=3, kinds=("c", "sse", "hip"), dense=False):
    ew, eh = size
    assert ew % 16 == 0 and eh % 16 == 0, "the enhancement layer's size must be a multiple of 16 (the base layer's of 8)"
    stats = ps.DENSE_QP22 if dense else NATURAL
    common = dict(gop="random_access", nframes=frames, gop_size=8, seed=7, log2_ctb=6)
    pb = ps.StreamParams(width=ew // 2, height=eh // 2, **common, **stats)
    pe = ps.StreamParams(width=ew, height=eh, tmvp=1, **common, **stats)
    t = time.perf_counter()
    aus, gen_bl, gen_el = ps.generate_shvc(pb, pe)
    tgen = time.perf_counter() - t
    res = dict(workload=f"synthetic two-layer random-access stream, base layer {ew // 2}x{eh // 2}, enhancement layer {ew}x{eh} (x2), 8 bit, {frames} access units x "
                        f"{passes} passes, {sum(map(len, aus)) // len(aus)} bytes / access unit ({'qp22-like density' if dense else 'encoder-like CU statistics'})",
               generate_s=round(tgen, 1), unit="access units/s (one decoding thread per layer; the enhancement layer's pictures per second)")
    ref = None
    for kind in kinds:
        if not ps.have(kind):
            res[kind] = None
            continue
        bl, el = ps.decode_stream_shvc(kind, aus)
        if ref is None:
            ref = (bl, el)
            want = (gen_bl, gen_el)           # the first decoder is checked against the generator's own reconstruction, the others against the first
        else:
            want = ref
        exact = all(len(got) == len(w) and all(np.array_equal(x, y) for fa, fb in zip(got, w) for x, y in zip(fa, fb)) for got, w in ((bl, want[0]), (el, want[1])))
        dt, n = timed(kind, aus, passes)
        res[kind] = dict(aus_per_s=round(len(aus) * passes / dt, 2), pictures=n, exact=bool(exact))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1088")
    ap.add_argument("--frames", type=int, default=17)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--kinds", default="c,sse,hip")
    ap.add_argument("--dense", action="store_true", help="qp22-like residual density in both layers instead of the encoder-like statistics")
    a = ap.parse_args()
    print(json.dumps(run(tuple(map(int, a.size.split("x"))), a.frames, a.passes, a.kinds.split(","), a.dense)))


if __name__ == "__main__":
    main()
