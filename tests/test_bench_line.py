"""The line the driver parses stays small, valid and complete (a 24 KB line was not parsed).

bench.py prints ``compact_line(full)`` as its LAST stdout line and writes the full record to a sidecar file; these tests feed
``compact_line`` the full records of earlier rounds (profiles/r0*_bench_line.json) and a deliberately bloated one."""
import copy
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("alo_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")
ROOF = ("bound", "achieved", "peak", "unit", "frac", "traffic")


def _canned(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["r05_bench_line.json", "r05_bench_line_box2.json", "r04_bench_line.json", "r03_bench_line.json"])
def test_compact_line_of_earlier_records(bench, name):
    full = _canned(name)
    text = bench.compact_line(full)
    assert "\n" not in text and len(text.encode()) < 4096
    line = json.loads(text)
    for key in CONTRACT:
        assert key in line, key
    for key in ROOF:
        assert key in line["roofline"], key
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    assert line["roofline"]["frac"] == full["roofline"]["frac"]
    assert line["roofline"]["bound"] in ("hbm", "mfma") and line["roofline"]["unit"] in ("GB/s", "TFLOP/s")
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind"}
    assert line["config"]["workload"] and "model" not in line["config"]
    # no prose: every string value in the line is short
    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(s) for s in strings(line)) <= 120


def test_compact_line_survives_a_bloated_record(bench):
    full = copy.deepcopy(_canned("r05_bench_line.json"))
    full["roofline"]["kernel"] = "k" * 5000
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["kernels"] = {f"kernel_{i}": {"note": "x" * 200} for i in range(500)}
    for leg in ("raft", "train", "fp32", "panoptic", "trained_like"):
        full[leg]["note"] = "n" * 10000
    text = bench.compact_line(full)
    assert len(text.encode()) < 4096
    line = json.loads(text)
    assert line["value"] == full["value"] and line["roofline"]["frac"] == full["roofline"]["frac"]
    assert line["raft"]["value"] == full["raft"]["value"] and line["train"]["frac"] == full["train"]["roofline"]["frac"]


def test_compact_line_carries_failed_legs_and_rejects_nan(bench):
    full = copy.deepcopy(_canned("r05_bench_line.json"))
    full["raft"] = {"error": "RuntimeError: " + "e" * 400}
    line = json.loads(bench.compact_line(full))
    assert set(line["raft"]) == {"error"} and len(line["raft"]["error"]) <= 120
    full["value"] = float("nan")
    with pytest.raises(ValueError):
        bench.compact_line(full)


def test_emit_writes_sidecar_and_prints_compact_line_last(bench, tmp_path, capsys):
    full = copy.deepcopy(_canned("r05_bench_line.json"))

    class A:
        detail_out = str(tmp_path / "detail.json")
        print_detail = False

    bench.emit(full, A)
    out = capsys.readouterr().out.strip().splitlines()
    line = json.loads(out[-1])
    assert len(out[-1]) < 4096 and line["value"] == full["value"]
    with open(A.detail_out) as f:
        detail = json.load(f)
    assert "kernels" in detail and "micro" in detail and detail["value"] == full["value"]
