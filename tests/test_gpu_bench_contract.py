"""bench.py prints exactly one JSON line with the fields the driver reads (run as a child process, small batch)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True,
                         timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_default_workload_line():
    d = run_bench("--steps", "3", "--warmup", "1", "--batch", "96")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["unit"] == "audio-seconds/s" and d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - 96 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.02
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s", "GFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_cnn_workload_line():
    d = run_bench("--workload", "cfg4", "--steps", "1", "--warmup", "1", "--batch", "2", "--no-cpu-baseline")
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["kernel"] == "k_cnn_forward"
    assert 0 < d["roofline"]["frac"] < 1 and d["value"] > 0
