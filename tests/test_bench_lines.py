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


def test_round_4_driver_line_carries_the_new_blocks():
    """profiles/r04_bench_line_driver_command.json: the kernels with the most device time as first-class entries (rate consistent with
    the 157.3 TFLOP/s divisor, shares of the step), the busy fraction from the event brackets, the high-convergence leg (>= 80 % of the
    rays converging), configs[2], the alternative matrix mode — and the 2-rank line the launcher produced on a GPU box."""
    d = json.loads((PROFILES / "r04_bench_line_driver_command.json").read_text().strip().splitlines()[-1])
    fc = d["roofline"]["first_class"]
    names = [e["kernel"] for e in fc]
    assert any("gemm_nt_narrow_kernel" in n for n in names) and any("gemm_tn_occ_kernel" in n for n in names)
    for e in fc:
        assert e["frac"] == pytest.approx(e["achieved"] / 157.3, abs=2e-3) and 0 <= e["share_of_step"] < 0.6 and e["launches"] > 0
    assert fc == sorted(fc, key=lambda e: -e["share_of_step"])
    b = d["busy_fraction_timed_region"]
    assert 0.3 < b["mfma_launches_over_4_gflop"] < 1.0 and b["union_s"] <= b["elapsed_s"]
    hc = d["high_convergence"]
    assert hc["rays_converged_fraction"] >= 0.8 and hc["value"] > 0 and "error" not in hc
    assert d["config2"]["rays_converged_fraction"] == 1.0 and d["alt_mode"]["gemm_mode"] == "bf16x6"
    assert "r04_pmc_loop.json" in d["roofline"]["traffic_source"] or "r03_pmc_loop.json" in d["roofline"]["traffic_source"]
    two = json.loads((PROFILES / "r04_bench_line_2ranks_one_gpu_launcher.json").read_text().strip().splitlines()[-1])
    assert two["n_gpus"] == 2 and two["config"]["replicas_bit_identical"] is True and len(two["config"]["per_rank_ms_per_step"]) == 2


def test_round_6_driver_line_is_on_the_frozen_scene():
    """profiles/r06_bench_line_driver_command.json: the workload comes from the committed scene file and is named at the top level
    (`matrix_tflop_per_step`, `mc_vertices`, `rays_per_iter`), `roofline.traffic` and `alg_bytes` describe the SAME launches, the
    over-unity sampler fraction is gone, the experimental matrix mode is off the line, configs[2] carries its re-mesh split."""
    d = json.loads((PROFILES / "r06_bench_line_driver_command.json").read_text().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "scene", "matrix_tflop_per_step", "mc_vertices", "rays_per_iter"):
        assert key in d, key
    assert d["scene"]["file"] == "configs/synthetic/bench_scene_v1.pt" and (PROFILES.parent / d["scene"]["file"]).is_file()
    assert d["dtype"] == "f32" and d["vs_baseline"] is None and d["n_gpus"] == 1 and "alt_mode" not in d
    assert 7.0 < d["matrix_tflop_per_step"] < 9.0 and len(d["mc_vertices"]) == 2 and d["rays_per_iter"] > 5000
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 0.02 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=2e-3)
    assert r["traffic"] == r["traffic_large_launches"]["traffic_bytes_per_launch"]          # same launch subset as alg_bytes
    assert 1.0 < r["traffic"] / r["alg_bytes"] < 2.5 and "r06_pmc_loop.json" in r["traffic_source"]
    assert all("frac_survey_8d" not in k for k in d["hbm_kernels"])
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and 0 < c["value"] < d["value"]
    split = d["config2"]["split"]
    for key in ("remesh_ms", "plain_ms", "pyramid_query_ms", "seg3d_bookkeeping_ms", "marching_cubes_ms", "mesh_handover_ms"):
        assert split[key] >= 0, key
    assert abs(split["remesh_ms"] + split["plain_ms"] - d["config2"]["ms_per_step"]) < 0.5
    assert d["remesh"]["remesh_steps_in_timed_region"] >= 1 and "device_allocations" in d["remesh"]
    two = json.loads((PROFILES / "r06_bench_line_2ranks_one_gpu.json").read_text().strip().splitlines()[-1])
    assert two["n_gpus"] == 2 and two["config"]["replicas_bit_identical"] is True and two["scene"]["file"] == d["scene"]["file"]
