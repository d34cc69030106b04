"""The bench line's contract, checked on the CPU twice: (a) against what bench.py's own build_line() assembles from STUBBED timings --
a change to bench.py that drops or renames a key, or breaks the arithmetic between the fields, fails here --, and (b) against the line
committed with the round's profiles (`python bench.py` on an MI355X), which also carries cpu_baseline and other_configs (they need
the GPU and the oracle).  The keys the driver and the review read, their units and their internal consistency."""
import importlib.util
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r06_a_bench.json")


def _bench_module():
    spec = importlib.util.spec_from_file_location("rf_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _stub_line():
    """build_line() on the numbers of a typical 1024^3 run: 20 steps in 77.2 ms, the merged launch at 0.170 ms"""
    b = _bench_module()
    return b.build_line((1024, 1024, 1024), 20, 5, 1, 0.0772, 77.0, 2.3137, np.array([1.05, 1.65, 1.33, 0.0186, 0.066]), 0.170, 15, 16, 64)


@pytest.fixture(scope="module", params=["built from stubbed timings", "committed"])
def line(request):
    if request.param == "committed":
        return json.load(open(LINE))
    out = _stub_line()
    # what main() adds after build_line(): stand-ins with the right shape (the real ones need a GPU and the oracle)
    out["cpu_baseline"] = {"value": 32.5, "unit": "Mcells/s", "cores": 1, "kind": "port", "sample": "stub", "rms": 2.3137536}
    out["speedup_vs_cpu_baseline"] = round(out["value"] / 32.5, 1)
    return out


def test_build_line_follows_its_inputs():
    """the arithmetic of the line: value and ms_per_step from the wall clock, the roofline object from the merged launch's duration,
    and -- for a shape without merged launches -- from the slowest pass"""
    b = _bench_module()
    out = _stub_line()
    assert out["ms_per_step"] == pytest.approx(3.86, rel=1e-6) and out["value"] == pytest.approx(1024 ** 3 / 3.86e-3 / 1e6, rel=1e-4)
    sweep = 8.0 * 1024 * 1024 * 513
    r = out["roofline"]
    assert "yz_merged_kernel" in r["kernel"] and r["launches_per_realisation"] == 15 and r["algorithmic_bytes_per_launch"] == 4 * sweep / 16
    assert r["achieved"] == pytest.approx(4 * sweep / 16 / 0.170e-3 / 1e9, rel=1e-3)
    assert r["unmerged_dominant_pass"]["kernel"].startswith("y pass") and out["pipeline"]["yz_slabs"] == 16
    plain = b.build_line((512, 512, 512), 10, 2, 1, 0.0054, 5.3, 2.2, np.array([0.14, 0.17, 0.19, 0.01, 0.02]), None, 0, 2, 256)
    assert plain["roofline"]["kernel"].startswith("z pass") and plain["roofline"]["launches_per_realisation"] == 2
    assert plain["roofline"]["achieved"] == pytest.approx(2 * 8.0 * 512 * 512 * 257 / 0.19e-3 / 1e9, rel=1e-3)
    assert plain["roofline"]["traffic"] is None and "1024" in plain["roofline"]["traffic_source"]      # (the committed PMC passes are 1024^3)


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


def test_other_configs_carry_every_baseline_configuration():
    line = json.load(open(LINE))
    o = line["other_configs"]
    for needle in ("512^3 f32 single realisation", "rng='reference'", "1024^3 f64", "1024^3 f64 + lognormal", "2048^3 f32 on one GPU",
                   "per-rank compute", "exchange stand-in", "numpy array on the host"):
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
        # the direct exchange's stand-in (no copy kernel at all): far below the copy stand-in's
        for k in ("direct exchange stand-in, one stream", "direct exchange stand-in, storing y pass on the exchange stream"):
            assert 0 < e[k]["pipelined_ms_per_realisation"] < e["128 workgroups"]["pipelined_ms_per_realisation"]
