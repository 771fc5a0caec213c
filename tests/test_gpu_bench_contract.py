"""bench.py prints ONE JSON line with the contract's keys (plus roofline / cpu_baseline), measured on the HIP path."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", *extra],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_json_contract(gpu):
    d = _run("--no-cpu-baseline")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "maps/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - d["config"]["global_batch"]) < 1e-6
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["unit"] in ("TFLOP/s", "GB/s") and r["launches"] >= 2
    assert 0.05 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 1e9


def test_bench_cpu_baseline_leg(gpu):
    d = _run("--workload", "cfg1")
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "maps/s" and c["sample"]
