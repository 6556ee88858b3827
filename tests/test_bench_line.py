"""The printed bench line stays small enough for the driver to read (a 25.9 KB line was cut off in the driver's stdout
tail and did not parse).  CPU test on a canned detailed result: round 5's own 25.9 KB dict."""
import json
import os

import bench_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _canned():
    with open(os.path.join(ROOT, "profiles", "r05z_bench_line.json")) as fh:
        return json.load(fh)


def test_printed_line_is_small_and_complete():
    line = _canned()
    assert len(json.dumps(line)) > 20000                     # the canned dict IS the one that broke the parse
    s = bench_line.dumps(line)
    assert "\n" not in s and len(s.encode()) <= bench_line.LINE_LIMIT <= 6144
    out = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert abs(out["value"] - line["value"]) <= 1e-4 * line["value"]
    assert len(out["config"]["workload"]) <= bench_line.STR_LIMIT
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(out["cpu_baseline"])
    assert out["calibration"]["value_s"] > 0 and "per_unit_ms" not in out["calibration"]
    assert set(out["configs"]) == {"cifar", "church", "sd"}
    assert "dropped_for_size" not in out


def test_oversized_values_cannot_break_the_bound():
    line = _canned()
    line["config"]["workload"] = "x" * 5000
    line["roofline"]["kernel"] = "k" * 5000
    line["calibration"]["stages"].update({"stage_%d" % i: float(i) for i in range(400)})     # forces the drop path
    s = bench_line.dumps(line)
    assert len(s.encode()) <= bench_line.LINE_LIMIT
    out = json.loads(s)
    assert out["value"] and out["roofline"]["frac"]


def test_partial_result_still_prints():
    s = bench_line.dumps({"metric": "m", "value": 1.0, "unit": "images/sec", "calibration": {"error": "boom " * 200}})
    out = json.loads(s)
    assert out["value"] == 1.0 and len(out["calibration"]["error"]) <= bench_line.STR_LIMIT


def test_bench_prints_through_the_compact_form():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "print(json.dumps(line" not in src and "bench_line.dumps(line)" in src
