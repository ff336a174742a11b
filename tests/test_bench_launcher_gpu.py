"""-m gpu: `python bench.py --gpus 2` from a bare shell - no torch.distributed.run - on the one GPU of the test box: bench.py starts one rank per
"GPU" itself (both on GPU 0: gloo for the barrier, the native transport's sockets wire for the pictures, which the line says), runs the kernel
bench on every rank and the frame-parallel decoder (BASELINE config 5's structure: owner = decoding-order index mod world, planes and motion
fields broadcast in bands of CTU rows) in child processes, and prints ONE JSON line.  This is the command path the driver's scaling run takes on a
multi-GPU node (there: one GPU per rank, RCCL for both), exercised every round."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_bench_gpus_2_from_a_bare_shell():
    from oracle import pystream as ps
    assert ps.have("hip"), "oracle/_ref/libopenhevc_hip.so missing: run __graft_entry__.build() where /root/reference exists"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--blocks", "65536", "--no-zscan",
                        "--check-blocks", "64", "--frames-size", "1920x1080", "--frames-bit-depth", "10", "--frames-pictures", "9", "--frames-steps", "2"],
                       env=env, capture_output=True, text=True, timeout=540, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line on stdout, got {len(lines)}: {r.stdout[-2000:]}"
    assert len(lines[0]) < 8192                            # the line the driver parses stays short (bench.compact_line); the rest is in the detail file
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["checked"] is True
    assert d["config"]["one_gpu"] is True                  # two ranks, one device: said in the line
    assert d["roofline"]["frac"] > 0
    sf = d["summary"]["frames"]                            # the compact row of the line ...
    assert "error" not in sf, sf
    assert sf["n_gpus"] == 2 and sf["bit_exact"] is True and sf["wire_ranks"] == 2 and sf["fps"] > 0 and sf["idr_segments"]["bit_exact"] is True
    full = json.load(open(os.path.join(ROOT, d["detail"])))   # ... and the full object, next to bench.py
    assert full["value"] == d["value"] and full["roofline"] == {**d["roofline"], "traffic_source": full["roofline"]["traffic_source"]}
    f = full["frames"]
    assert "error" not in f, f
    assert f["n_gpus"] == 2 and f["scaling"] == "strong"
    assert f["bit_exact"] is True and f["one_rank"]["pictures_checked"] == 9 and f["one_rank"]["pictures_differing"] == 0
    assert f["wire_ranks"] == 2
    assert f["exchange"]["pictures_exchanged"] > 0 and f["exchange"]["bands_imported"] > 0 and f["exchange"]["bytes_per_exchanged_picture"] > 1920 * 1080 * 2
    assert f["fps"] > 0 and f["one_rank"]["fps"] > 0
    g = f["idr_segments"]                                   # the same stream, ownership per IDR segment: nothing crosses the wire
    assert g["bit_exact"] is True and g["pictures_exchanged"] == 0 and g["pictures_checked"] == 18 and g["fps"] > 0
