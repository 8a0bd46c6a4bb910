"""bench.py's stdout contract line: BENCH_r05.json came back `parsed: null` because the one line had
grown to 20 KB.  The line is now built by benchlib.common.contract_line from the detail object; here a round-5-sized detail
object (the committed profiles/bench_r05zd.json, the very line the driver could not parse) goes through it."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from benchlib.common import CONTRACT_LINE_MAX_BYTES, contract_line  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _round5_detail():
    return json.load(open(os.path.join(ROOT, "profiles", "bench_r05zd.json")))


def test_contract_line_is_small_and_complete():
    out = _round5_detail()
    assert len(json.dumps(out)) > 15000          # the object that broke the driver's parser
    s = contract_line(out)
    assert "\n" not in s and len(s.encode()) < CONTRACT_LINE_MAX_BYTES <= 4096
    d = json.loads(s)
    for k in CONTRACT:
        assert k in d, k
    assert d["value"] == round(out["value"], 4) and d["ms_per_step"] == round(out["ms_per_step"], 4)
    assert set(d["config"]) >= {"workload", "path", "library"} and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "algorithmic_bytes_per_launch", "traffic_source"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert set(rf["traffic_source"]) == {"file", "stale"}
    cb = d["cpu_baseline"]
    assert set(cb) >= {"value", "unit", "cores", "kind", "dense_value", "sample"} and cb["kind"] in ("port", "reference")
    assert "verified_layers" not in d and "families" not in rf and "yardstick" not in rf


def test_contract_line_survives_a_bloated_detail_object():
    """whatever a later round adds to the detail object, the line stays under the limit (optional blocks are dropped first)"""
    out = _round5_detail()
    out["stages"]["f32_split"] = {"planes%d_%d_ms" % (i, j): 1.2345678 for i in range(40) for j in range(20)}
    out["config"]["path"] = "x" * 5000
    out["cpu_baseline"]["sample"] = "y" * 5000
    s = contract_line(out)
    assert len(s.encode()) < CONTRACT_LINE_MAX_BYTES
    d = json.loads(s)
    for k in CONTRACT + ("roofline", "cpu_baseline"):
        assert k in d, k
