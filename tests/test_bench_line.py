"""The line bench.py hands the driver: ONE JSON object on the last line of stdout, small enough to survive the driver's
8 KB stdout tail (round 5's grew past 20 KB and was recorded as `parsed: null`), carrying the contract's keys, the
dominant kernel's roofline and the CPU baseline; everything else goes to the details file.  The reference's own benchmark
prints one number (internal_tools/hw_decoder_bench.py:651)."""
import copy
import io
import json
import os
import subprocess
import sys
from contextlib import redirect_stdout

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic")


def _full_line():
    # a complete details object of an earlier round (every leg present: the largest input the compaction ever sees)
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))


def _check(line, text):
    assert len(text) < bench.COMPACT_LINE_LIMIT < 6000
    for k in REQUIRED:
        assert k in line, k
    for k in ROOFLINE:
        assert k in line["roofline"], k
    assert line["roofline"]["bound"] in ("hbm", "mfma")
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-6
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert isinstance(line["config"]["workload"], str) and "model" not in line["config"]
    # the dominant kernel's launch cannot last longer than the step it is part of ... with one batch in flight (several
    # batches overlap in the timed region: a launch next to its neighbours may)
    alone = line["roofline"].get("alone_ms") or line["roofline"]["avg_ms"]
    assert alone <= line["ms_per_step"] * 1.15
    assert all(isinstance(v, (int, float, bool, str, type(None))) for v in line["config"].values())


def test_compact_line_of_a_complete_run_is_small_and_complete():
    full = _full_line()
    assert len(json.dumps(full)) > 15000                     # (the thing that broke round 5)
    line = bench.compact_line(full, "bench_details.json")
    _check(line, json.dumps(line))
    assert line["value"] == pytest.approx(full["value"], rel=1e-5)
    assert line["config"]["e2e_images_per_s"] == pytest.approx(full["e2e_pipeline"]["value"], rel=1e-5)
    assert line["config"]["resident_indexed_images_per_s"] == pytest.approx(
        full["config"]["pipeline"]["resident_indexed"]["value"], rel=1e-5)


def test_compact_line_stays_small_whatever_the_details_grow_to():
    full = copy.deepcopy(_full_line())
    full["config"]["workload"] = "x" * 50000
    full["cpu_baseline"]["sample"] = "y" * 50000
    full["config"]["pipeline"]["note"] = "z" * 100000
    for i in range(500):
        full["config"][f"new_nested_{i}"] = {"a": list(range(50))}
        full[f"new_leg_{i}"] = {"value": 1.0, "note": "n" * 1000}
    line = bench.compact_line(full, "bench_details.json")
    _check(line, json.dumps(line))


def test_emit_prints_the_compact_line_last_and_writes_the_details(tmp_path, monkeypatch):
    monkeypatch.setenv("BENCH_DETAILS", str(tmp_path / "d.json"))
    buf = io.StringIO()
    with redirect_stdout(buf):
        print("bench: some progress line")
        bench.emit(_full_line())
    last = buf.getvalue().strip().splitlines()[-1]
    _check(json.loads(last), last)
    details = json.load(open(tmp_path / "d.json"))
    assert "per_kernel" in details["roofline"] and "e2e_pipeline" in details


def test_default_command_runs_no_side_legs():
    # the driver's command is `python bench.py --gpus 1 --steps K --warmup W`: the informational legs are opt-in
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'ap.add_argument("--side-legs", action="store_true"' in src
    for leg in ("bench_heavy_aug), (", "iterator_leg(args", "pillow_baseline(enc_all", "cache_mb=1024"):
        at = src.index(leg, src.index("def main()"))
        assert "args.side_legs" in src[src.rfind("\n        if ", 0, at) - 400:at] or "args.side_legs" in src[at - 900:at], leg


@pytest.mark.gpu
def test_driver_command_last_stdout_line_parses(tmp_path):
    env = dict(os.environ, BENCH_DETAILS=str(tmp_path / "d.json"))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    last = res.stdout.strip().splitlines()[-1]
    line = json.loads(last)
    _check(line, last)
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["warmup"] == 2
    assert line["config"]["e2e_images_per_s"] > 0 and line["config"]["resident_indexed_images_per_s"] > 0
    assert line["config"]["resident_set_MB"] > 0
    tail = res.stdout[-8192:]                                   # what the driver keeps
    assert json.loads(tail.strip().splitlines()[-1]) == line
