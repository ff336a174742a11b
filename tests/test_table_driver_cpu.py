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


@pytest.mark.parametrize("bd,W,H", [(8, 256, 136), (10, 192, 128)])
def test_driver_on_reference_tables_equals_oracle_stream(oracle, ref, bd, W, H):
    rng = np.random.default_rng(31 + bd)
    dt = np.uint16 if bd > 8 else np.uint8
    dims = X.chroma_dims(W, H)
    refs = [[rng.integers(0, 1 << bd, size=d).astype(dt) for d in dims] for _ in range(2)]
    cur0 = [rng.integers(0, 1 << bd, size=d).astype(dt) for d in dims]
    ops, fops = S.gen_frame_ops(rng, W, H, bd, intra_frac=0.3, pcm_frac=0.05)
    assert any(o["t"] == "pcm" for o in ops)
    want = X.run_oracle(oracle, po, bd, W, H, [p.copy() for p in cur0], refs, ops, fops)
    got = [p.copy() for p in cur0]
    rc = X.drive_tables(ref.lib, bd, W, H, got, refs, X.encode_driver_ops(ops, fops))
    assert rc == 0
    for c in range(3):
        bad = np.argwhere(got[c] != want[c])
        assert bad.size == 0, f"plane {c}: {len(bad)} mismatches, first {bad[:4].tolist()}"
