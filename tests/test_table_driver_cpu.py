"""CPU: the miniature front-end (oracle/table_driver.c) driving the REFERENCE's own tables must reproduce what the
C restatement produces for the same op stream in decode order -- pins the driver (used by the drop-in test on the GPU)
and the oracle at whole-picture level, including edge emulation and put_pcm through the reference's bit reader."""
import os
import sys

import numpy as np
import pytest

from oracle import pyoracle as po
import stream_exec as X

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import synth_stream as S  # noqa: E402


def pcm_ops_for(rng, bd):
    out = []
    for (c_idx, x, y, log2) in [(0, 8, 8, 3), (1, 4, 4, 2), (2, 4, 4, 2), (0, 64, 32, 4)]:
        pcm_bd = int(rng.integers(4, bd + 1))
        n = 1 << log2
        out.append(dict(c_idx=c_idx, x=x, y=y, log2=log2, pcm_bd=pcm_bd, samples=rng.integers(0, 1 << pcm_bd, size=(n, n))))
    return out


@pytest.mark.parametrize("bd,W,H", [(8, 256, 136), (10, 192, 128)])
def test_driver_on_reference_tables_equals_oracle_stream(oracle, ref, bd, W, H):
    rng = np.random.default_rng(31 + bd)
    dt = np.uint16 if bd > 8 else np.uint8
    dims = X.chroma_dims(W, H)
    refs = [[rng.integers(0, 1 << bd, size=d).astype(dt) for d in dims] for _ in range(2)]
    cur0 = [rng.integers(0, 1 << bd, size=d).astype(dt) for d in dims]
    ops, fops = S.gen_frame_ops(rng, W, H, bd, intra_frac=0.3)
    pcm = pcm_ops_for(rng, bd)
    # oracle, decode order; PCM blocks replace their area before the in-loop filters
    want = X.run_oracle(oracle, po, bd, W, H, [p.copy() for p in cur0], refs, ops, [])
    for p in pcm:
        n = 1 << p["log2"]
        want[p["c_idx"]][p["y"]:p["y"] + n, p["x"]:p["x"] + n] = (p["samples"] << (bd - p["pcm_bd"])).astype(dt)
    want = X.run_oracle(oracle, po, bd, W, H, want, refs, [], fops)
    got = [p.copy() for p in cur0]
    rc = X.drive_tables(ref.lib, bd, W, H, got, refs, X.encode_driver_ops(ops, fops, pcm))
    assert rc == 0
    for c in range(3):
        bad = np.argwhere(got[c] != want[c])
        assert bad.size == 0, f"plane {c}: {len(bad)} mismatches, first {bad[:4].tolist()}"
