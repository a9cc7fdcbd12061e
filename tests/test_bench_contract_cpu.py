"""The bench line contract, checked on the lines committed under profiles/ (no GPU needed):
one JSON object with the driver's fields, a `roofline` object for the dominant kernel and a
`cpu_baseline` object; `config` names a workload (no model keys)."""
import glob
import json
import os

import pytest

from conftest import ROOT

LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r01f_bench_*.json")))


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_bench_lines_follow_the_contract(path):
    text = open(path).read().strip()
    assert "\n" not in text, "one JSON line"
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "steps/s" and d["higher_is_better"] is True and d["scaling"] == "strong"
    assert d["vs_baseline"] is None            # BASELINE.md publishes no number for this metric
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-3)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=1e-3)
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-3)
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"]
    cpu = d.get("cpu_baseline")
    if "under_rocprof" not in path:            # the profiled runs skip the CPU leg
        assert cpu and cpu["kind"] == "port" and cpu["unit"] == "steps/s" and cpu["cores"] >= 1 and cpu["value"] > 0
        assert "sample" in cpu


def test_default_line_carries_both_rooflines():
    d = json.loads(open(os.path.join(ROOT, "profiles", "r01f_bench_default_iterative_schur.json")).read())
    assert "kSx" in d["roofline"]["kernel"] and "kJtJx" in d["roofline_jtjx"]["kernel"]
    assert d["config"]["solver"].startswith("ITERATIVE_SCHUR")
    assert d["extra"]["cgnr"]["steps_per_s"] > 0
