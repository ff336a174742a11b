"""Fuzzing of the two-layer (SHVC) drop-in: random layer geometries, ratios and coding tools -> synthesiser (oracle/pystream.py: generate_shvc)
-> the reference's pair of decoders on its own C tables vs the same pair with the gfx950 back end (FUZZ_BACKEND=hip) or with the device code on
the host emulator (FUZZ_BACKEND=hipemu, the default here: no GPU needed).  Prints every failing parameter set as JSON.

    python tools/fuzz_shvc.py [seconds] [seed]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                      # noqa: E402
from oracle import pystream as ps       # noqa: E402
from oracle import pyoracle as po       # noqa: E402

REFLIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libhevcref.so")

BACKEND = os.environ.get("FUZZ_BACKEND", "hipemu")


def random_layers(rng):
    log2_ctb_b, log2_ctb_e = int(rng.choice([4, 5, 6])), int(rng.choice([4, 5, 6]))
    kind = str(rng.choice(["x2", "x2", "x1_5", "general", "general", "snr"]))
    if kind == "x2":
        bw, bh = int(rng.integers(6, 30)) * 8, int(rng.integers(6, 20)) * 8
        ew, eh = 2 * bw, 2 * bh
    elif kind == "x1_5":
        bw, bh = int(rng.integers(3, 15)) * 16, int(rng.integers(3, 10)) * 16
        ew, eh = bw * 3 // 2, bh * 3 // 2
    elif kind == "snr":
        bw, bh = int(rng.integers(6, 40)) * 8, int(rng.integers(6, 26)) * 8
        ew, eh = bw, bh
    else:       # any ratio between 1 and 3, not the same in both directions
        bw, bh = int(rng.integers(6, 30)) * 8, int(rng.integers(6, 20)) * 8
        ew = int(round(bw * float(rng.uniform(1.0, 3.0)) / 8)) * 8
        eh = int(round(bh * float(rng.uniform(1.0, 3.0)) / 8)) * 8
    common = dict(gop=str(rng.choice(["lowdelay_p", "lowdelay_b", "random_access", "intra"])), nframes=int(rng.integers(2, 7)),
                  gop_size=int(rng.choice([4, 8])), seed=int(rng.integers(1, 1 << 30)))

    def layer(w, h, log2_ctb):
        kw = dict(width=w, height=h, log2_ctb=log2_ctb, log2_max_tb=min(5, log2_ctb), amp=int(rng.integers(0, 2)), sao=int(rng.integers(0, 4) != 0),
                  strong_intra_smoothing=int(rng.integers(0, 2)), tmvp=int(rng.integers(0, 2)), sign_hiding=int(rng.integers(0, 2)),
                  init_qp=int(rng.integers(14, 46)), constrained_intra=int(rng.integers(0, 5) == 0), transform_skip=int(rng.integers(0, 2)),
                  cu_qp_delta_depth=int(rng.integers(-1, 3)), weighted_pred=int(rng.integers(0, 3) == 0), weighted_bipred=int(rng.integers(0, 3) == 0),
                  deblock_control=int(rng.integers(0, 2)), max_merge_cand=int(rng.integers(1, 6)), tu_depth_inter=int(rng.integers(0, 3)),
                  tu_depth_intra=int(rng.integers(0, 3)), log2_parallel_merge_level=int(rng.integers(2, log2_ctb + 1)))
        if rng.integers(0, 5) == 0:
            kw["pcm"] = int(rng.integers(5, 9))
            kw["pcm_log2_max"] = min(5, log2_ctb)
        if rng.integers(0, 6) == 0:
            kw["transquant_bypass"] = 1
        ctb_w, ctb_h = -(-w >> log2_ctb), -(-h >> log2_ctb)
        mode = int(rng.integers(0, 4))
        if mode == 1 and ctb_h > 1:
            kw["wpp"] = 1
        elif mode == 2 and ctb_w >= 2 and ctb_h >= 2:
            kw["tiles"] = (int(rng.integers(1, min(3, ctb_w) + 1)), int(rng.integers(1, min(3, ctb_h) + 1)))
            if kw["tiles"] == (1, 1):
                kw.pop("tiles")
        r = int(rng.integers(0, 4))
        if r == 0:
            kw["probs"] = dict(ps.DENSE_QP22["probs"])
        elif r == 1:      # encoder-like: mostly skipped / merged, i.e. mostly inter-layer or temporal prediction without a residual
            kw["probs"] = dict(pred_mode=0.05, skip=0.55, merge_flag=0.7, split_cu=0.35, rqt_root_cbf=0.45, cbf_luma=0.5, cbf_chroma=0.25)
        return kw

    kb, ke = layer(bw, bh, log2_ctb_b), layer(ew, eh, log2_ctb_e)
    # several slices: the base layer at least as many as the enhancement layer (set_refindex_data, hevc_refs.c:373-394, reads the base-layer
    # picture's reference list of the same slice index)
    nb = int(rng.integers(1, 5)) if "tiles" not in kb else 1
    ne = int(rng.integers(1, nb + 1)) if "tiles" not in ke else 1
    if nb > 1:
        kb["slices_per_picture"] = nb
    if ne > 1:
        ke["slices_per_picture"] = ne
        ke["dependent_slices"] = int(rng.integers(0, 2))
    # (x1.5 with phase alignment through the block slots: the reference reads scratch rows it did not prepare)
    return dict(common, **kb), dict(common, **ke), int(rng.integers(0, 2)) if kind != "x1_5" else 0, kind


def same(a, b):
    return len(a) == len(b) and all(np.array_equal(x, y) for fa, fb in zip(a, b) for x, y in zip(fa, fb))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time()
    n = bad = gen_fail = outside = 0
    kinds = {}
    while time.time() - t0 < budget:
        kb, ke, pa, kind = random_layers(rng)
        # geometries for which the reference's own CTB-by-CTB resampling reads the base-layer frame buffer's edge or rows of its scratch buffer it
        # did not prepare (oracle/pyoracle.py): its output then depends on the order CTBs were resampled in and on what the buffers held before
        if kind in ("general", "x1_5", "x2") and any(po.shvc_reference_not_a_function_of_its_inputs(REFLIB, ke["width"], ke["height"], kb["width"], kb["height"], pa, ke["log2_ctb"])):
            outside += 1
            continue
        try:
            aus, gen_bl, gen_el = ps.generate_shvc(ps.StreamParams(**kb), ps.StreamParams(**ke), pa)
            ref_bl, ref_el = ps.decode_stream_shvc("c", aus)
            if not (same(ref_bl, gen_bl) and same(ref_el, gen_el)):
                print("GENERATOR != REFERENCE", json.dumps([kb, ke, pa]), flush=True)
                gen_fail += 1
                continue
        except Exception as e:      # an illegal random combination: not a back-end problem    # noqa: BLE001
            if os.environ.get("FUZZ_VERBOSE"):
                print("GEN FAIL", type(e).__name__, e, json.dumps([kb, ke, pa]), flush=True)
            gen_fail += 1
            continue
        n += 1
        kinds[kind] = kinds.get(kind, 0) + 1
        if os.environ.get("FUZZ_VERBOSE"):
            print("TRY", kind, json.dumps([kb, ke, pa]), flush=True)
        try:
            bl, el = ps.decode_stream_shvc(BACKEND, aus)
            ok = same(bl, ref_bl) and same(el, ref_el)
            why = "" if ok else ("base layer differs" if not same(bl, ref_bl) else "enhancement layer differs")
        except Exception as e:      # noqa: BLE001
            ok, why = False, f"{type(e).__name__}: {e}"
        if not ok:
            bad += 1
            print("FAIL", why, json.dumps([kb, ke, pa]), flush=True)
    print(json.dumps(dict(backend=BACKEND, streams=n, failures=bad, rejected_by_the_synthesiser=gen_fail, reference_not_a_function_of_its_inputs=outside, by_ratio=kinds, seconds=round(time.time() - t0, 1))))


if __name__ == "__main__":
    main()
