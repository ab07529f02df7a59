"""The committed bench lines under profiles/ keep the driver's contract: one JSON object with the metric / value / timing
fields, the `roofline` and `cpu_baseline` objects, a workload `config` without model keys.  (Host-only check of the files the
docs cite; the numbers themselves come from the GPU runs.)"""
import json
from pathlib import Path

import pytest

PROFILES = Path(__file__).resolve().parent.parent / "profiles"
LINES = sorted(PROFILES.glob("r02_bench_line_v[67]_*.json"))


def test_final_lines_are_committed():
    names = {p.name for p in LINES}
    assert {"r02_bench_line_v6_final.json", "r02_bench_line_v6_traced_command.json",
            "r02_bench_line_v6_driver_command.json"} <= names


@pytest.mark.parametrize("path", LINES, ids=lambda p: p.name)
def test_bench_line_contract(path):
    d = json.loads(path.read_text())
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["unit"] == "iters/s"
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 0.02 * d["value"]            # one GPU: iterations/s = 1000 / ms per step
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == pytest.approx(157.3)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=2e-3)
    assert r["traffic"] is None or r["traffic"] > 1e8
    if "cpu_baseline" in d and d["cpu_baseline"]:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "iters/s" and 0 < c["value"] < d["value"]
    assert d["remesh"]["remesh_steps_in_timed_region"] >= 1                      # a re-mesh is always inside the timed steps


def test_whole_step_matrix_rate_is_a_function_of_the_line():
    """bench.whole_step_matrix_rate on the committed driver-command line: Σ MFMA FLOP of a step over the step time."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", PROFILES.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    line = json.loads((PROFILES / "r02_bench_line_v7_driver_command.json").read_text().strip().splitlines()[-1])
    w = bench.whole_step_matrix_rate(line["roofline"], line["steps"], line["ms_per_step"])
    assert 6.0 < w["matrix_tflop_per_step"] < 10.0 and 0.3 < w["frac_of_f32_mfma_peak"] < 0.6
    assert w["achieved"] == round(w["matrix_tflop_per_step"] / line["ms_per_step"] * 1e3, 2) or abs(
        w["achieved"] - w["matrix_tflop_per_step"] / line["ms_per_step"] * 1e3) < 0.05
    print(w)
