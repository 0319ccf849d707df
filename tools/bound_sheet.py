#!/usr/bin/env python3
"""Recompute the bound sheet of DESIGN.md (section 6) from the committed measurements:
    tools/bound_sheet.py profiles/r03_a_pmc_sq_cfg3.txt profiles/r03_a_bench_cfg3_line.json [profiles/r03_a_traffic_cfg3.json]
Per kernel of the cfg3 step: measured time, VALU issue time (wave-instructions x the micro-benchmark rates: 2 cycles per
f32 instruction, 2.42 ns per float64 one), LDS time (SQ_LDS_IDX_ACTIVE / CUs / clock), memory time (PMC bytes / measured
streaming rates), and the step's roofline fraction. Works on the files of the two-kernel route (rounds 1-2: K1
k_erb_filterbank + K2 k_envelope) and of the spectral route (round 3 on: KS k_spectral_envelope + its two pre-kernels)."""
import json, re, sys

SIMDS, CUS = 1024, 256
F64_NS = 2.42          # ns per v_fma_f64 wave-instruction per SIMD at two waves per SIMD (tools/ubench/fma64_operands.hip)
F32_CYC = 2.0          # cycles per f32 VALU wave-instruction (SIMD-32)
HBM_MIX, HBM_WR = 5.27e12, 6.2e12   # bytes/s: K2's read+write mix, pure writes (tools/ubench/stream_rw.hip)
PEAK = 8e12
KERNELS = (("K1", "k_erb_filterbank", "k_erb_filterbank"), ("K2", "k_envelope", "k_envelope<float, 13>"),
           ("KS", "k_spectral_envelope", "k_spectral_envelope<13"), ("KX", "k_utterance_spectrum", "k_utterance_spectrum"),
           ("KT", "k_tail_state", "k_tail_state"))

pmc = {}
cur = None
for line in open(sys.argv[1]):
    if not line.startswith(" "):
        cur = next((tag for tag, _, pat in KERNELS if pat in line), None)
        continue
    m = re.match(r"\s+(\S+)\s+mean\s+([\d.]+)", line)
    if cur and m:
        pmc.setdefault(cur, {})[m.group(1)] = float(m.group(2))
line = json.load(open(sys.argv[2]))
traffic = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
C, samples = 128, 1000 * 16000
sc = C * samples
valu_total = 0.0
for tag, name, _ in KERNELS:
    if tag not in pmc or name not in line["kernels"]:
        continue
    p = pmc[tag]
    ms = line["kernels"][name]["ms_per_step"]
    if ms < 0.02:      # the two-kernel route's launches that only skip utterances served by the spectral kernel
        print(f"{tag} {name}: {ms * 1e3:.0f} us per step (launched to serve utterances the accuracy guard sends back: none here)")
        continue
    clock = p["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) if "GRBM_GUI_ACTIVE" in p and ms > 0.3 else 1.98e9
    insts = p["SQ_INSTS_VALU"]
    f64 = 14.3 * sc / 64 if tag == "K1" else 0.0     # (KT / KX: float64 too, but a few % of the step; priced as f32 here)
    valu_ms = (f64 * F64_NS * 1e-9 / SIMDS + (insts - f64) * F32_CYC / SIMDS / 1.98e9) * 1e3
    valu_total += valu_ms
    lds_ms = p.get("SQ_LDS_IDX_ACTIVE", 0.0) / CUS / 1.98e9 * 1e3
    byts = traffic[name]["hbm_bytes_per_launch"] if traffic and name in traffic else None
    hbm_ms = byts / (HBM_MIX if tag == "K2" else HBM_WR) * 1e3 if byts else float("nan")
    waves = p.get("SQ_WAVE_CYCLES", 0)
    print(f"{tag} {name}: measured {ms:.2f} ms (profiled clock {clock / 1e9:.2f} GHz) | VALU issue {valu_ms:.2f} ms "
          f"({insts / 1e6:.0f} M wave-instr, {f64 / 1e6:.0f} M float64) | LDS {lds_ms:.2f} ms | memory side {hbm_ms:.2f} ms "
          f"({(byts or 0) / 1e9:.2f} GB)")
    if waves:
        print(f"     wave time: {100 * p['SQ_ACTIVE_INST_ANY'] / waves:.0f} % issuing, {100 * p['SQ_WAIT_INST_ANY'] / waves:.0f} % stalled "
              f"at issue ({100 * p.get('SQ_WAIT_INST_LDS', 0) / waves:.1f} % on the LDS queue), {100 * p['SQ_WAIT_ANY'] / waves:.0f} % parked")
need = (2 + 8 * C) * samples
print(f"step: {line['ms_per_step']:.2f} ms, {need / 1e9:.2f} GB required -> {need / (line['ms_per_step'] * 1e-3) / PEAK:.3f} of {PEAK / 1e12:.0f} TB/s")
print(f"VALU issue alone ({valu_total:.2f} ms over the kernels above) caps the step at "
      f"{need / (valu_total * 1e-3) / PEAK:.2f} of the HBM roofline")
