#!/usr/bin/env python3
"""Lane-level numpy emulation of tu_idct32_mfma (openhevc_amd/csrc/tu_kernels.hip): the same operand construction, byte splits,
constants and register orders, run through a model of v_mfma_i32_32x32x32_i8's fragment layout, and compared with the oracle's
32x32 inverse transform.  Checks the index algebra of the kernel without a GPU (what it cannot check is the layout model itself:
A[m = lane & 31][k = 16 * (lane >> 5) + byte], B[k][n = lane & 31] likewise, D[m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][n = lane & 31]).
    python tools/emulate_idct32_mfma.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import pyoracle as po


def cos_mag(m):
    t = [64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0]
    return t[m]


def dct32(r, c):
    m = (r * (2 * c + 1)) & 127
    return cos_mag(m) if m <= 32 else -cos_mag(64 - m) if m <= 64 else -cos_mag(m - 64) if m <= 96 else cos_mag(128 - m)


T = np.array([[dct32(k, y) for y in range(32)] for k in range(32)], dtype=np.int64)     # T[k][y]
S = T.sum(axis=0)                                                                        # sum over k of T[k][y]


def mfma(a, b, c):
    """a, b: [64][16] int8 (lane, byte); c: [64][16] int32 (lane, reg)."""
    A = np.zeros((32, 32), np.int64); B = np.zeros((32, 32), np.int64)
    for l in range(64):
        for j in range(16):
            A[l & 31, 16 * (l >> 5) + j] = a[l, j]
            B[16 * (l >> 5) + j, l & 31] = b[l, j]
    D = A @ B
    out = c.astype(np.int64).copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    assert np.all(np.abs(out) < 2 ** 31)
    return out


def k1(h, j):      # coefficient row held in byte j of half h of the pass-1 A operand (four transposing LDS reads of 4 rows each)
    return 16 * h + j


def k2(h, j):      # pass-1 output register j of half h is column ... of the block (the D layout)
    return (j & 3) + 8 * (j >> 2) + 4 * h


def split(v16):
    """int16 values [64][16] -> (hi, lo) int8 [64][16]: v = 256 * hi + (lo + 128)."""
    u = v16.astype(np.int64) & 0xffff
    hi = ((u >> 8) & 0xff).astype(np.uint8).view(np.int8)
    lo = ((u & 0xff) ^ 0x80).astype(np.uint8).view(np.int8)
    return hi, lo


def clip16(v):
    return np.clip(v, -32768, 32767)


def emulate(coeffs, bd):
    lane = np.arange(64)
    # transposing reads: lane l ends up with rows 16 (l >> 5) .. + 15 of column l & 31
    w = np.stack([coeffs[k1(lane >> 5, j), lane & 31] for j in range(16)], axis=1)           # [64][16] int16
    a_hi, a_lo = split(w)
    b1 = np.array([[T[k1(l >> 5, j), l & 31] for j in range(16)] for l in range(64)], dtype=np.int8)
    init1 = np.repeat((64 + 128 * S[lane & 31])[:, None], 16, axis=1)
    d_hi = mfma(a_hi, b1, np.zeros((64, 16), np.int64))
    d_lo = mfma(a_lo, b1, init1)
    p1 = clip16((d_hi * 256 + d_lo) >> 7)                                   # lane = y, reg r = column k2(h, r)
    b_hi, b_lo = split(p1)
    a2 = np.array([[T[k2(l >> 5, j), l & 31] for j in range(16)] for l in range(64)], dtype=np.int8)   # A2[m = x][k = c]
    shift2 = 20 - bd
    xs = np.array([[k2(l >> 5, r) for r in range(16)] for l in range(64)])   # output register r of half h = x
    init2 = (1 << (shift2 - 1)) + 128 * S[xs]
    e_hi = mfma(a2, b_hi, np.zeros((64, 16), np.int64))
    e_lo = mfma(a2, b_lo, init2)
    res = clip16((e_hi * 256 + e_lo) >> shift2)                             # lane = y (+ half), reg r = x = k2(h, r)
    out = np.zeros((32, 32), np.int64)
    for l in range(64):
        for r in range(16):
            out[l & 31, k2(l >> 5, r)] = res[l, r]
    return out


def main():
    oracle = po.load("oracle") if hasattr(po, "load") else None
    if oracle is None:
        oracle = po.Oracle(os.path.join(os.path.dirname(po.__file__), "libhevc_oracle.so"), "ohor_")
    rng = np.random.default_rng(7)
    for it in range(12):
        bd = int(rng.choice([8, 9, 10, 12]))
        if it < 3:
            c = rng.integers(-32768, 32768, size=(32, 32)).astype(np.int16)
        elif it < 6:
            c = np.zeros((32, 32), np.int16); c[rng.integers(0, 32), rng.integers(0, 32)] = rng.integers(-32768, 32768)
        else:
            c = rng.integers(-1024, 1024, size=(32, 32)).astype(np.int16)
        want = oracle.tu_residual(bd, po.TU_IDCT, 5, c)
        got = emulate(c, bd)
        assert np.array_equal(want, got), (it, bd, np.argwhere(want != got)[:4])
    print("emulation matches the oracle on 12 blocks")


if __name__ == "__main__":
    main()
