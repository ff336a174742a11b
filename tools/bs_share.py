#!/usr/bin/env python3
"""What the boundary strengths cost the front end (second half: would deriving them on the device pay?).

The reference derives them per transform unit on the host (ff_hevc_deblocking_boundary_strengths, hevc_filter.c:805-941; boundary_strength :584-700).
This tool decodes the same synthetic 1080p streams with two builds of the no-pixels decoder (oracle/_ref/libopenhevc_null*.so: every table
slot empty) - one as it is, one with that function turned into a no-op - so the difference is exactly the host time the boundary strengths
cost: the MOST a device-side derivation could give back.  Against it stands what the device would need instead: the motion field
(MvField, 24 bytes per 4x4 in the reference's layout; 12 + 1 bytes packed) copied out of the decoder's arrays per picture and uploaded -
timed here as the packing loop a recorder would run.  CPU only; prints one JSON line per stream."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pystream as ps       # noqa: E402

NATURAL = dict(init_qp=32, probs=dict(pred_mode=0.03, skip=0.55, merge_flag=0.7, split_cu=0.3, rqt_root_cbf=0.45, cbf_luma=0.5, cbf_chroma=0.25,
                                      split_transform=0.25, sig_coeff=0.35, last_x=0.5, last_y=0.5))


def best_of(kind, aus, n=5):
    best = None
    for _ in range(n):
        t = time.perf_counter()
        ps.decode_stream(kind, aus)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    return best


def main():
    assert ps.have("null") and ps.have("null_nobs"), "make -C oracle nobs"
    W, H, N = 1920, 1080, 17
    # what a recorder would do per picture instead: pack mv[2] + the low 16 bits of poc[2] (12 bytes) and pred_flag (1 byte) per 4x4
    mvf = np.zeros((W // 4) * (H // 4), dtype=np.dtype([("mv", "<i2", 4), ("poc", "<i4", 2), ("pred_flag", "<u4"), ("ref_idx", "u1", 2), ("pad", "u1", 2)]))
    t = time.perf_counter()
    for _ in range(20):
        packed = np.empty((mvf.size, 6), np.int16)
        packed[:, :4] = mvf["mv"]
        packed[:, 4:] = mvf["poc"].astype(np.int16)
        flags = mvf["pred_flag"].astype(np.uint8)
    pack_ms = (time.perf_counter() - t) / 20 * 1e3
    t = time.perf_counter()
    for _ in range(20):
        raw = mvf.tobytes()
    copy_ms = (time.perf_counter() - t) / 20 * 1e3
    for name, extra in (("natural", NATURAL), ("flat", {}), ("qp22", ps.DENSE_QP22)):
        kw = dict(gop="random_access", nframes=N, seed=7, width=W, height=H, log2_ctb=6, bit_depth=8)
        kw.update(extra)
        aus, _ = ps.generate(ps.StreamParams(**kw))
        a, b = best_of("null", aus), best_of("null_nobs", aus)
        print(json.dumps(dict(stream=name, pictures=N, front_end_ms_per_picture=round(a / N * 1e3, 3), without_boundary_strengths_ms=round(b / N * 1e3, 3),
                              boundary_strengths_ms_per_picture=round((a - b) / N * 1e3, 3), share=round((a - b) / a, 4),
                              device_side_would_add=dict(motion_field_bytes_reference_layout=int(mvf.nbytes), packed_bytes=int(packed.nbytes + flags.nbytes),
                                                         host_copy_of_the_reference_layout_ms=round(copy_ms, 3), host_packing_numpy_ms=round(pack_ms, 3),
                                                         pcie_ms_at_50GBps=round(mvf.nbytes / 50e9 * 1e3, 3)))), flush=True)


if __name__ == "__main__":
    main()
