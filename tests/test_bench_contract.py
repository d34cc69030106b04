"""The bench line's contract, checked on the CPU against the line committed with the round's profiles (profiles/r05_c_bench.json =
`python bench.py --steps 20 --warmup 5` at HEAD on an MI355X): the keys the driver and the review read, their units and their internal
consistency.  (bench.py itself needs a GPU; this keeps a refactor from silently dropping or renaming a key.)"""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r05_c_bench.json")


@pytest.fixture(scope="module")
def line():
    return json.load(open(LINE))


def test_top_level_contract(line):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["unit"] == "Mcells/s" and line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and line["n_gpus"] == 1
    assert "workload" in line["config"] and "model" not in line["config"]
    nx, ny, nz = line["config"]["grid"]
    # value = cells * steps / wall, ms_per_step = wall / steps
    assert abs(line["value"] - nx * ny * nz / (line["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * line["value"]


def test_roofline_object(line):
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_ms", "hbm_bytes_est", "frac_hbm_est"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3
    # achieved = algorithmic bytes per launch / mean launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_ms"] * 1e-3) / 1e9) <= 1e-3 * r["achieved"]
    # the counter traffic of the committed PMC passes is per launch and close to the algorithmic bytes (no wasted re-reads)
    assert r["traffic"] is None or 0.9 <= r["traffic"] / r["algorithmic_bytes_per_launch"] <= 1.15
    # the HBM-pin estimate: half of the merged launch's four slab sweeps, and it says that it is an estimate
    assert abs(r["hbm_bytes_est"] - r["algorithmic_bytes_per_launch"] / 2) <= 1 and "ESTIMATE" in r["hbm_bytes_est_note"]
    assert abs(r["frac_hbm_est"] - r["hbm_bytes_est"] / (r["avg_ms"] * 1e-3) / 1e9 / r["peak"]) <= 1e-3
    p = line["pipeline"]
    nx, ny, nz = line["config"]["grid"]
    sweep = 8.0 * nx * ny * (nz // 2 + 1)
    assert p["hbm_bytes_est_per_realisation"] == 3 * sweep
    assert abs(p["frac_of_hbm_peak"] - 5 * sweep / (line["ms_per_step"] * 1e-3) / 1e9 / 8000.0) <= 2e-3


def test_cpu_baseline_object(line):
    c = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["unit"] == "Mcells/s" and c["kind"] in ("port", "reference") and c["cores"] == 1
    assert line["speedup_vs_cpu_baseline"] == pytest.approx(line["value"] / c["value"], rel=1e-3)
    assert abs(c["rms"] - 2.3137536) < 1e-6            # the reference's own rms for this workload


def test_other_configs_carry_every_baseline_configuration(line):
    o = line["other_configs"]
    for needle in ("512^3 f32 single realisation", "rng='reference'", "1024^3 f64", "1024^3 f64 + lognormal", "2048^3 f32 on one GPU",
                   "per-rank compute", "exchange stand-in"):
        assert any(needle in k for k in o), needle
    for k, v in o.items():
        if isinstance(v, dict) and "ms" in v and "kernel_ms" in v:
            assert "roofline" in v and {"bound", "achieved", "peak", "unit", "frac"} <= set(v["roofline"]), k
    st = next(v for k, v in o.items() if "exchange stand-in" in k)
    for rank in ("rank 0", "rank 3"):
        e = st[rank]
        assert e["standin_bytes_read"] == e["standin_bytes_written"] == 7.0 / 8.0 * 8.0 * 2048 * 2048 * 1025 / 8.0
        for w in ("16 workgroups", "32 workgroups"):
            assert e[w]["pipelined_ms_per_realisation"] > e["forward_plus_backward_ms"] > 0
            assert e[w]["slowdown_vs_forward_plus_backward"] == pytest.approx(e[w]["pipelined_ms_per_realisation"] / e["forward_plus_backward_ms"], rel=2e-3)
