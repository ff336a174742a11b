#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own compiled C (oracle/_ref/libhevcref.so, built by oracle/Makefile from
/root/reference).  The reference ships no golden vectors; these fixtures freeze its behaviour on seeded
inputs so that the oracle stays pinned where /root/reference is absent.  Re-run only in an environment that has the
reference:  python tests/golden/make_golden.py
Each file holds the call arguments and the reference's output; tests/test_oracle_golden.py replays the arguments through
oracle/liboracle.so (the C restatement) and compares bit for bit."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import pyoracle as po  # noqa: E402


def pixdt(bd):
    return np.uint16 if bd > 8 else np.uint8


def gen_cases(lib, rng):
    """Yields (family, args dict, expected array).  `lib` is any OracleLib; the generator only defines inputs."""
    cases = []
    for bd in (8, 10):
        # residual kinds
        for log2 in (2, 3, 4, 5):
            n = 1 << log2
            kinds = [po.TU_IDCT, po.TU_DC, po.TU_SKIP, po.TU_SKIP_RDPCM_H, po.TU_SKIP_RDPCM_V, po.TU_BYPASS_RDPCM_H, po.TU_BYPASS_RDPCM_V]
            if log2 == 2:
                kinds.append(po.TU_DST4)
            for kind in kinds:
                for amp in (1 << 15, 700):
                    c = rng.integers(-amp, amp, size=(n, n)).astype(np.int16)
                    cases.append(("tu", dict(bd=bd, kind=kind, log2=log2, coeffs=c, col_limit=n)))
            c = rng.integers(-900, 900, size=(n, n)).astype(np.int16)
            for cl in (4, 8, 12, 24):
                if cl < n:
                    cases.append(("tu", dict(bd=bd, kind=po.TU_IDCT, log2=log2, coeffs=c, col_limit=cl)))
        # MC
        refp = rng.integers(0, 1 << bd, size=(48, 96)).astype(pixdt(bd))
        for luma in (1, 0):
            for w in ([4, 8, 12, 16, 24, 32, 48, 64] if luma else [2, 4, 6, 8, 12, 16, 24, 32]):
                for (mx, my) in [(0, 0), (2, 0), (0, 1), (3, 3)]:
                    src2 = rng.integers(-8000, 16000, size=(8, 64)).astype(np.int16)
                    for variant in range(5):
                        cases.append(("mc", dict(bd=bd, luma=luma, variant=variant, ref=refp, sx=8, sy=8, w=w, h=8, mx=mx, my=my, src2=src2,
                                                 denom=int(rng.integers(0, 8)), wx0=int(rng.integers(-128, 128)), wx1=int(rng.integers(-128, 128)),
                                                 ox0=int(rng.integers(-128, 128)), ox1=int(rng.integers(-128, 128)))))
        # deblock
        for it in range(40):
            base = int(rng.integers(0, 1 << bd)); amp = int(rng.choice([1, 2, 4, 16])) << (bd - 8)
            plane = np.clip(base + rng.integers(-amp, amp + 1, size=(16, 16)), 0, (1 << bd) - 1).astype(pixdt(bd))
            for vert in (0, 1):
                for chroma in (0, 1):
                    cases.append(("dbk", dict(bd=bd, vert=vert, chroma=chroma, plane=plane, beta=int(rng.integers(0, 65)),
                                              tc=[int(rng.integers(0, 25)), int(rng.integers(0, 25))],
                                              no_p=[int(rng.random() < 0.1), int(rng.random() < 0.1)], no_q=[int(rng.random() < 0.1), int(rng.random() < 0.1)])))
        # SAO
        for it in range(16):
            src = rng.integers(0, 1 << bd, size=(34, 34)).astype(pixdt(bd))
            if it % 2:
                src = ((src >> (bd - 3)) + (1 << (bd - 1))).astype(src.dtype)
            ov = np.concatenate([[0], rng.integers(-31, 32, size=4) << (bd - 8)]).astype(np.int16)
            cases.append(("sao_band", dict(bd=bd, src=src, ov=ov, bp=int(rng.integers(0, 32)))))
            for eo in range(4):
                cases.append(("sao_edge", dict(bd=bd, src=src, ov=ov, eo=eo, restore=it % 2, borders=[int(rng.random() < 0.3) for _ in range(4)],
                                               ve=[int(rng.random() < 0.3) for _ in range(2)], he=[int(rng.random() < 0.3) for _ in range(2)],
                                               de=[int(rng.random() < 0.3) for _ in range(4)])))
        # intra predictors + full intra_pred
        for log2 in (2, 3, 4, 5):
            n = 1 << log2
            for mode in range(35):
                cases.append(("pred", dict(bd=bd, log2=log2, mode=mode, c_idx=mode % 2, top=rng.integers(0, 1 << bd, size=2 * n + 9),
                                           left=rng.integers(0, 1 << bd, size=2 * n + 9))))
        for it in range(60):
            log2 = int(rng.integers(2, 6)); n = 1 << log2; c_idx = int(rng.integers(0, 3))
            nl = n << (1 if c_idx else 0)
            x0 = int(rng.integers(0, (136 - nl) // nl + 1)) * nl; y0 = int(rng.integers(0, (72 - nl) // nl + 1)) * nl
            cands = [int(rng.random() < 0.7) for _ in range(5)]
            if x0 == 0: cands[0] = cands[1] = cands[2] = 0
            if y0 == 0: cands[2] = cands[3] = cands[4] = 0
            if x0 + nl >= 136: cands[4] = 0
            if y0 + nl >= 72: cands[0] = 0
            planes = [rng.integers(0, 1 << bd, size=(80, 144)).astype(pixdt(bd)) for _ in range(3)]
            cases.append(("intra", dict(bd=bd, planes=planes, x0=x0, y0=y0, log2=log2, c_idx=c_idx, mode=int(rng.integers(0, 35)), cands=cands,
                                        strong=int(rng.random() < 0.7), dis=int(rng.random() < 0.1), ctb=int(rng.choice([4, 5, 6])))))
    return cases


def run_case(lib, fam, a):
    bd = a["bd"]
    if fam == "tu":
        return lib.tu_residual(bd, a["kind"], a["log2"], a["coeffs"], a["col_limit"])
    if fam == "mc":
        return lib.mc(bd, a["luma"], a["variant"], a["ref"], a["sx"], a["sy"], a["w"], a["h"], a["mx"], a["my"], src2=a["src2"],
                      denom=a["denom"], wx0=a["wx0"], wx1=a["wx1"], ox0=a["ox0"], ox1=a["ox1"])
    if fam == "dbk":
        p = a["plane"].copy()
        x, y = (8, 4) if a["vert"] else (4, 8)
        if a["chroma"]:
            lib.deblock_chroma(bd, a["vert"], p, x, y, a["tc"], a["no_p"], a["no_q"])
        else:
            lib.deblock_luma(bd, a["vert"], p, x, y, a["beta"], a["tc"], a["no_p"], a["no_q"])
        return p
    if fam == "sao_band":
        d = np.zeros_like(a["src"]); lib.sao_band(bd, d, a["src"], 1, 1, 32, 32, a["ov"], a["bp"]); return d
    if fam == "sao_edge":
        d = np.zeros_like(a["src"])
        lib.sao_edge(bd, a["restore"], d, a["src"], 1, 1, 32, 32, a["ov"], a["eo"], a["borders"], a["ve"], a["he"], a["de"]); return d
    if fam == "pred":
        return lib.pred(bd, a["log2"], a["mode"], a["top"], a["left"], a["c_idx"])
    if fam == "intra":
        pl = [p.copy() for p in a["planes"]]
        lib.intra_pred(bd, pl, 136, 72, a["x0"], a["y0"], a["log2"], a["c_idx"], a["mode"], a["cands"], chroma_format_idc=1,
                       strong=a["strong"], smoothing_disabled=a["dis"], log2_ctb_size=a["ctb"], log2_min_tb_size=2)
        return pl[a["c_idx"]]
    raise ValueError(fam)


def digest(arr):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def main():
    ref = po.load("ref")
    if ref is None:
        raise SystemExit("oracle/_ref/libhevcref.so missing: build it with `make -C oracle ref` (needs /root/reference)")
    cases = gen_cases(ref, np.random.default_rng(20260924))
    fams = {}
    for fam, args in cases:
        fams.setdefault(fam, []).append(digest(run_case(ref, fam, args)))
    # inputs are regenerated from the seed by the test (same generator); only the reference's output digests are stored
    np.savez_compressed(os.path.join(HERE, "reference_digests.npz"), **{k: np.array(v) for k, v in fams.items()})
    # plus a few full vectors for eyeballing / other tools
    full = {}
    for fam in ("tu", "mc", "intra"):
        for k, (f, a) in enumerate([c for c in cases if c[0] == fam][:3]):
            full[f"{fam}{k}_out"] = run_case(ref, f, a)
    np.savez_compressed(os.path.join(HERE, "reference_samples.npz"), **full)
    print({k: len(v) for k, v in fams.items()})


if __name__ == "__main__":
    main()
