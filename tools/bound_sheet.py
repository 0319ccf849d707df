#!/usr/bin/env python3
"""Recompute the bound sheet of DESIGN.md section 6a from the committed measurements:
    tools/bound_sheet.py profiles/r02_c_pmc_sq_cfg3.txt profiles/r02_c_bench_cfg3_line.json [profiles/r02_c_traffic_cfg3.json]
Per kernel: measured time, VALU issue time (float64 / f32 wave-instructions x the micro-benchmark rates), LDS time
(SQ_LDS_IDX_ACTIVE / CUs / clock), HBM time (PMC bytes / measured streaming rates), and the step's roofline fraction."""
import json, re, sys

SIMDS, CUS = 1024, 256
F64_NS = 2.42          # ns per v_fma_f64 wave-instruction per SIMD at two waves per SIMD (tools/ubench/fma64_operands.hip)
F32_CYC = 2.0          # cycles per f32 VALU wave-instruction (SIMD-32)
HBM_MIX, HBM_WR = 5.27e12, 6.2e12   # bytes/s: K2's read+write mix, pure writes (tools/ubench/stream_rw.hip)
PEAK = 8e12

pmc = {}
cur = None
for line in open(sys.argv[1]):
    if not line.startswith(" "):
        cur = "K1" if "k_erb_filterbank" in line else "K2" if "k_envelope<float, 13>" in line else None
        continue
    m = re.match(r"\s+(\S+)\s+mean\s+([\d.]+)", line)
    if cur and m:
        pmc.setdefault(cur, {})[m.group(1)] = float(m.group(2))
line = json.load(open(sys.argv[2]))
traffic = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
C, samples = 128, 1000 * 16000
sc = C * samples
for k, name in (("K1", "k_erb_filterbank"), ("K2", "k_envelope")):
    p = pmc[k]
    ms = line["kernels"][name]["ms_per_step"]
    clock = p["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) if "GRBM_GUI_ACTIVE" in p else 1.98e9
    insts = p["SQ_INSTS_VALU"]
    f64 = 14.3 * sc / 64 if k == "K1" else 0.0
    valu_ms = (f64 * F64_NS * 1e-9 / SIMDS + (insts - f64) * F32_CYC / SIMDS / 1.98e9) * 1e3
    lds_ms = p["SQ_LDS_IDX_ACTIVE"] / CUS / 1.98e9 * 1e3
    byts = traffic[name]["hbm_bytes_per_launch"] if traffic else None
    hbm_ms = byts / (HBM_WR if k == "K1" else HBM_MIX) * 1e3 if byts else float("nan")
    waves = p.get("SQ_WAVE_CYCLES", 0)
    print(f"{k} {name}: measured {ms:.2f} ms (profiled clock {clock / 1e9:.2f} GHz) | VALU issue {valu_ms:.2f} ms "
          f"({insts / 1e6:.0f} M wave-instr, {f64 / 1e6:.0f} M float64) | LDS {lds_ms:.2f} ms | HBM {hbm_ms:.2f} ms "
          f"({(byts or 0) / 1e9:.2f} GB)")
    if waves:
        print(f"     wave time: {100 * p['SQ_ACTIVE_INST_ANY'] / waves:.0f} % issuing, {100 * p['SQ_WAIT_INST_ANY'] / waves:.0f} % stalled "
              f"at issue ({100 * p.get('SQ_WAIT_INST_LDS', 0) / waves:.1f} % on the LDS queue), {100 * p['SQ_WAIT_ANY'] / waves:.0f} % parked")
need = (2 + 8 * C) * samples
print(f"step: {line['ms_per_step']:.2f} ms, {need / 1e9:.2f} GB required -> {need / (line['ms_per_step'] * 1e-3) / PEAK:.3f} of {PEAK / 1e12:.0f} TB/s")
k1f = 14.3 * sc / 64 * F64_NS * 1e-9 / SIMDS * 1e3
k2v = pmc["K2"]["SQ_INSTS_VALU"] * F32_CYC / SIMDS / 1.98e9 * 1e3
print(f"VALU issue alone (K1 float64 {k1f:.2f} ms + K2 f32 {k2v:.2f} ms = {k1f + k2v:.2f} ms) caps the step at "
      f"{need / ((k1f + k2v) * 1e-3) / PEAK:.2f} of the HBM roofline")
