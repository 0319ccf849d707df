"""The committed profile summaries stay readable by the tools that DESIGN.md quotes (no GPU needed)."""
import glob
import json
import os
import subprocess
import sys

from conftest import ROOT


def newest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert files, pattern
    return files[-1]


def test_bound_sheet_runs_on_the_committed_files():
    pmc, line, traffic = newest("r*_pmc_sq_cfg3.txt"), newest("r*_bench_cfg3_line.json"), newest("r*_traffic_cfg3.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bound_sheet.py"), pmc, line, traffic],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "of 8 TB/s" in out.stdout
    assert ("KS k_spectral_envelope" in out.stdout) or ("K1 k_erb_filterbank" in out.stdout and "K2 k_envelope" in out.stdout)


def test_default_line_keeps_the_contract_fields():
    d = json.load(open(newest("r*_bench_default_line.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]   # measured bytes never below the required ones
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    t = json.load(open(newest("r*_traffic_cfg3.json")))
    assert r["traffic"] is None or t[r["kernel"]]["hbm_bytes_per_launch"] == r["traffic"]
