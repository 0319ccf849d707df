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


def test_the_committed_traffic_counters_belong_to_the_kernels_in_the_tree():
    """bench.py quotes `roofline.traffic` (HBM bytes per launch from the rocprofv3 --pmc passes) only while the hash of
    f2cnn_amd/csrc/* equals the one stamped into the newest profiles/r*_traffic_cfg3.json - a kernel edit after the last profile
    round silently turns the field into null. Re-run tools/profile_round.sh and copy its traffic file when this fails."""
    sys.path.insert(0, ROOT)
    import bench
    t = json.load(open(newest("r*_traffic_cfg3.json")))
    assert t["lib_source_hash"] == bench.lib_source_hash()
    assert bench.measured_traffic("cfg3", 1000, 128, 16000, "f32")["k_spectral_envelope"] >= 8 * 128 * 16000 * 1000
