#!/usr/bin/env python3
"""Recovers the fragment layout of v_mfma_i32_32x32x32_i8 on the GPU at hand from one-hot probes (ohevc_debug_mfma_i8_probe) and
compares it with the model tu_idct32_mfma_kernel is written against:
    A[m = lane & 31][k <- (lane >> 5, byte)],  B[k][n = lane & 31] with the same k map,  D[m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][n = lane & 31].
Prints one JSON line; "model_ok": false comes with what was found instead.     python tools/probe_mfma_layout.py"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openhevc_amd import lib as L  # noqa: E402


def run(lib, a, b):
    n = a.shape[0]
    da, db = torch.from_numpy(a.view(np.uint8)).cuda(), torch.from_numpy(b.view(np.uint8)).cuda()
    dd = torch.zeros((n, 64, 16), dtype=torch.int32, device="cuda")
    lib.ohevc_debug_mfma_i8_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    rc = lib.ohevc_debug_mfma_i8_probe(da.data_ptr(), db.data_ptr(), dd.data_ptr(), n, None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return dd.cpu().numpy()


def main():
    lib = L.load_library()
    out = {"model_ok": True}
    # 1. where does A lane la put its row, and B lane lb its column?  a = one-hot byte 0 of lane la, b = all ones / vice versa
    a = np.zeros((64, 64, 16), np.int8); b = np.ones((64, 64, 16), np.int8)
    for la in range(64):
        a[la, la, 0] = 1
    d = run(lib, a, b)
    rows = []
    for la in range(64):
        cells = np.argwhere(d[la] != 0)          # (lane, reg) cells of row m(la): all 32 columns
        lanes = sorted(set(cells[:, 0].tolist())); regs = sorted(set(cells[:, 1].tolist()))
        rows.append((lanes, regs))
    # model: row m = la & 31 lives in lanes 32 * ((m >> 2) & 1) + n (n = 0..31), register r with (r & 3) + 8 (r >> 2) = m - 4 ((m >> 2) & 1)
    bad_rows = []
    for la in range(64):
        m = la & 31; half = (m >> 2) & 1
        r = (m & 3) + 4 * (m >> 3)
        want = (list(range(32 * half, 32 * half + 32)), [r])
        if rows[la] != want:
            bad_rows.append({"a_lane": la, "lanes": rows[la][0][:4] + ["..."] if len(rows[la][0]) > 4 else rows[la][0], "regs": rows[la][1]})
    if bad_rows:
        out["model_ok"] = False; out["d_rows_differ"] = bad_rows[:6]
    a = np.ones((64, 64, 16), np.int8); b = np.zeros((64, 64, 16), np.int8)
    for lb in range(64):
        b[lb, lb, 0] = 1
    d = run(lib, a, b)
    bad_cols = []
    for lb in range(64):
        lanes = sorted(set(np.argwhere(d[lb] != 0)[:, 0].tolist()))
        if lanes != [lb & 31, (lb & 31) + 32]:
            bad_cols.append({"b_lane": lb, "lanes": lanes[:6]})
    if bad_cols:
        out["model_ok"] = False; out["d_cols_differ"] = bad_cols[:6]
    # 2. k map: A (lane la in {0, 32}, byte ja) meets B (lane lb in {0, 32}, byte jb) iff same half and same byte
    probes = [(la, ja, lb, jb) for la in (0, 32) for ja in range(16) for lb in (0, 32) for jb in range(16)]
    a = np.zeros((len(probes), 64, 16), np.int8); b = np.zeros((len(probes), 64, 16), np.int8)
    for i, (la, ja, lb, jb) in enumerate(probes):
        a[i, la, ja] = 1; b[i, lb, jb] = 1
    d = run(lib, a, b)
    k_bad = []
    for i, (la, ja, lb, jb) in enumerate(probes):
        hit = bool(np.any(d[i] != 0))
        if hit != (la == lb and ja == jb):
            k_bad.append([la, ja, lb, jb, hit])
    if k_bad:
        out["model_ok"] = False; out["k_map_differs"] = k_bad[:12]; out["k_map_mismatches"] = len(k_bad)
    # 3. signedness: (-128) * (-128) and (-1) * 127
    a = np.zeros((2, 64, 16), np.int8); b = np.zeros((2, 64, 16), np.int8)
    a[0, 0, 0] = -128; b[0, 0, 0] = -128; a[1, 0, 0] = -1; b[1, 0, 0] = 127
    d = run(lib, a, b)
    out["signed_products"] = [int(d[0, 0, 0]), int(d[1, 0, 0])]
    if out["signed_products"] != [16384, -127]:
        out["model_ok"] = False
    # 4. ds_read_b64_tr_b16 with the kernel's addressing of a row-major [32][32] int16 image: lane l must receive rows 16 h + 4 t .. + 3
    #    of column l & 31 (h = l >> 5), t = 0..3
    lib.ohevc_debug_lds_tr16_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lane = np.arange(64); u = lane & 15
    base = (16 * (lane >> 5) + (u >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (u & 3)) * 2
    addr = np.stack([base + 256 * t for t in range(4)]).astype(np.int32)
    d_addr = torch.from_numpy(addr).cuda()
    d_out = torch.zeros((4, 64, 4), dtype=torch.int16, device="cuda")
    assert lib.ohevc_debug_lds_tr16_probe(d_addr.data_ptr(), d_out.data_ptr(), 4, None) == 0
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().astype(np.int64)
    want = np.array([[[(16 * (l >> 5) + 4 * t + e) * 32 + (l & 31) for e in range(4)] for l in range(64)] for t in range(4)])
    out["tr16_ok"] = bool(np.array_equal(got, want))
    if not out["tr16_ok"]:
        out["model_ok"] = False
        out["tr16_lanes_0_1_4_16_32"] = {str(l): got[0, l].tolist() for l in (0, 1, 4, 16, 32)}
        out["tr16_want_lanes_0_1_4_16_32"] = {str(l): want[0, l].tolist() for l in (0, 1, 4, 16, 32)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
