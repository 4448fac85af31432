"""bench.py prints ONE small JSON line on stdout (the driver parses it; round 4's 21 KB line was cut by the driver's
reader and left the round without a measurement) and writes everything else to a details file."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _strings(x):
    if isinstance(x, dict):
        for v in x.values():
            yield from _strings(v)
    elif isinstance(x, (list, tuple)):
        for v in x:
            yield from _strings(v)
    elif isinstance(x, str):
        yield x


@pytest.mark.parametrize("name", ["r04_bench_default_run.json", "r03_bench_default_run.json"])
def test_final_line_of_a_full_report_is_small(name):
    """the fullest reports this repository ever produced (17 and 21 KB) come out as a line below 3 KB with every
    contract key, roofline and cpu_baseline in it."""
    B = _bench()
    out = json.load(open(os.path.join(ROOT, "profiles", name)))
    text = B.final_line(out, "bench_details.json")
    assert "\n" not in text and len(text) < B.LINE_LIMIT <= 4096
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert line["value"] == pytest.approx(out["value"], rel=1e-6) and line["steps"] == out["steps"] and line["warmup"] == out["warmup"]
    assert all(len(s) <= 120 for s in _strings(line))
    assert set(line["config"]) == {"workload", "samples_per_step_per_gpu", "sharding", "steps_in_flight"}   # no model keys


def test_final_line_drops_extras_before_it_grows():
    """whatever the report holds, the line stays below the limit: the optional blocks go first"""
    B = _bench()
    out = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default_run.json")))
    out["configs"] = dict(("config%d" % i, out["configs"]["config3"]) for i in range(200))
    text = B.final_line(out, "x" * 500)
    assert len(text) < B.LINE_LIMIT
    line = json.loads(text)
    assert "configs" not in line and "roofline" in line and "cpu_baseline" in line


def test_final_line_of_an_n_gpu_report():
    """N > 1: no cpu_baseline (rank 0 at N = 1 only), the collective's figures and the measured strong-scaling row ride along"""
    B = _bench()
    out = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default_run.json")))
    del out["cpu_baseline"], out["api"]
    out["n_gpus"] = 8
    out["distributed"] = {"backend": "nccl", "world_size": 8, "devices": [["host", i, "uuid-%032d" % i] for i in range(8)],
                          "one_gpu_per_rank": True, "ranks_in_collective": 8}
    out["allgather"] = {"avg_ms": 0.123456789, "bytes_per_rank": 80000, "collective": "RCCL all_gather_into_tensor",
                        "backend": "nccl", "world_size": 8}
    out["strong_scaling"] = {"samples_total": 10000, "measured_on": "8 GPU(s)",
                             "config2": {"n8": out["strong_scaling"]["config2"]["n8"]},
                             "config3": {"n8": out["strong_scaling"]["config3"]["n8"]}}
    line = json.loads(B.final_line(out, "bench_details.json"))
    assert "cpu_baseline" not in line
    assert line["distributed"] == {"backend": "nccl", "world_size": 8, "one_gpu_per_rank": True, "ranks_in_collective": 8}
    assert line["allgather"]["avg_ms"] > 0 and line["strong_scaling"]["config2"]["n8"] > 0


def test_headline_quotes_the_longer_measurement_when_the_k_steps_are_a_blink():
    """20 steps of config 2 are 50 ms: where the same step repeated for a second says something else by more than 2 %, `value` is
    the longer measurement; K steps that are themselves half a second, or agree within 2 %, stay the value."""
    B = _bench()
    sus = {"value": 3.84e6, "ms_per_step": 2.604, "seconds": 1.02, "steps": 392}
    h = B.headline({"value": 4.02e6, "ms_per_step": 2.4876, "sustained": sus}, 20)
    assert h["value"] == 3.84e6 and h["ms_per_step"] == 2.604 and h["k_steps_value"] == 4.02e6 and h["value_source"].startswith("sustained")
    h = B.headline({"value": 3.90e6, "ms_per_step": 2.564, "sustained": sus}, 20)           # within 2 %
    assert h["value"] == 3.90e6 and h["value_source"].startswith("the 20 timed steps")
    h = B.headline({"value": 4.02e6, "ms_per_step": 2.4876, "sustained": sus}, 400)         # K steps of a second by themselves
    assert h["value"] == 4.02e6
    h = B.headline({"value": 4.02e6, "ms_per_step": 2.4876}, 20)                            # no sustained loop was run
    assert h["value"] == 4.02e6 and "sustained_value" not in h


def test_final_line_carries_the_target_shape_and_the_value_source():
    B = _bench()
    out = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default_run.json")))
    out.update(B.headline(out, out["steps"]))
    line = json.loads(B.final_line(out, "bench_details.json"))
    assert "value_source" in line and line["k_steps_value"] > 0
    c3 = line["configs"]["config3"]
    assert c3["kernel"].startswith("k_count_merged") and c3["cpu_baseline"]["kind"] == "port" and c3["cpu_baseline"]["cores"] >= 1
