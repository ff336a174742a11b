"""bench.py's last stdout line must stay short enough for the driver to parse (BENCH_r05.parsed was null: the line had grown to 22 KB).
The compact line is built here from a canned full result (tests/golden/bench_full_canned.json = the full object of a real run, round 5)."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline", "checked", "check", "summary", "detail")


def canned():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_canned.json")))


def text(line):
    return json.dumps(line, separators=(",", ":"))


def test_line_is_short_and_complete():
    full = canned()
    assert len(json.dumps(full)) > 16384                       # the canned object is the one the driver could not parse
    line = bench.compact_line(full)
    s = text(line)
    assert len(s) < bench.LINE_LIMIT <= 8192
    assert "\n" not in s
    for k in REQUIRED:
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert line["config"]["workload"].startswith("synthetic batched 32x32")
    # the headline numbers are the full object's, untouched
    for k in ("value", "ms_per_step", "steps", "warmup", "n_gpus"):
        assert line[k] == full[k]
    assert line["roofline"]["frac"] == full["roofline"]["frac"]


def test_summary_has_one_number_per_row():
    full = canned()
    sm = bench.compact_line(full)["summary"]
    assert set(sm["kernels_frac"]) == set(full["kernels"])
    for k, v in sm["kernels_frac"].items():
        assert abs(v - full["kernels"][k]["frac"]) < 1e-3
    for name, row in full["decode"]["streams"].items():
        r = sm["decode"][name]
        assert r["modes"] == ["1thread", "16frame_threads"]
        assert r["hip"] == [row["hip_1thread"]["fps"], row["hip_16frame_threads"]["fps"]]
        assert r["sse"] == [row["reference_sse_1thread"]["fps"], row["reference_sse_16frame_threads"]["fps"]]
        assert r["fe"] == [row["front_end_only_1thread"]["fps"], row["front_end_only_16frame_threads"]["fps"]]
        assert r["ok"] is True
    c4 = sm["decode"]["config4_4k_main10_wpp"]
    full4 = full["decode"]["sizes"]["config4_4k_main10_wpp"]
    assert c4["hip"] == [full4["hip_" + m]["fps"] for m in c4["modes"]]


def test_line_survives_growth_and_errors():
    full = canned()
    for i in range(400):                                        # a table ten times today's
        full["kernels"][f"extra_row_number_{i}_with_a_long_name_8bit"] = copy.deepcopy(full["kernels"]["sao_edge_luma_8bit"])
    line = bench.compact_line(full)
    assert len(text(line)) < bench.LINE_LIMIT
    assert "summary_dropped" in line and "kernels_frac" in line["summary_dropped"]
    for k in REQUIRED:
        assert k in line, k
    broken = canned()
    broken["decode"] = {"error": "RuntimeError: " + "x" * 5000}
    broken["kernels"] = {"error": "y" * 5000}
    broken["frames"] = {"error": "z" * 5000}
    line = bench.compact_line(broken)
    assert len(text(line)) < 4096
    minimal = {k: canned()[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                        "dtype", "data", "config", "roofline", "checked", "check")}
    assert "summary" in bench.compact_line(minimal)              # an N > 1 line: no cpu_baseline / kernels / decode


def test_emit_prints_the_compact_line_last(tmp_path, capsys, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(canned())
    lines = capsys.readouterr().out.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < bench.LINE_LIMIT
    assert json.loads(lines[0])["detail"] == bench.DETAIL_FILE
    assert json.load(open(tmp_path / bench.DETAIL_FILE))["decode"]["streams"]["natural"]["hip_1thread"]["fps"] > 0
