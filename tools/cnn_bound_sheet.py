#!/usr/bin/env python3
"""Bound sheet of the CNN kernels from committed measurements:
    tools/cnn_bound_sheet.py profiles/rNN_x_pmc_sq_k4.txt profiles/rNN_x_k4_kernel_stats.csv
(the SQ counter summary of `tools/k4_probe.py 14240` - three rocprofv3 --pmc passes, tools/profile_round.sh - and the kernel-trace
statistics of the same command). Per kernel: measured time, matrix-core time (SQ_VALU_MFMA_BUSY_CYCLES: 32 cycles per
v_mfma_f32_32x32x16_bf16 and SIMD) and its share, instructions per MFMA by kind, LDS array time and its bank-conflict share, and
where the waves' time goes (issuing / stalled at issue / parked at s_waitcnt or a barrier). SQ_* cycle counters are in units of
four cycles (MI355X_MICROARCH.md), SQ_VALU_MFMA_BUSY_CYCLES in cycles; SQ_INSTS_VALU includes the MFMAs."""
import csv, re, sys

SIMDS, CUS = 1024, 256
pmc, cur = {}, None
for line in open(sys.argv[1]):
    if not line.startswith(" "):
        cur = line.strip()[:60]
        continue
    m = re.match(r"\s+(\S+)\s+mean\s+([\d.]+)", line)
    if cur and m:
        pmc.setdefault(cur, {})[m.group(1)] = float(m.group(2))
times = {}
for r in csv.DictReader(open(sys.argv[2])):
    times[r["Name"]] = float(r["AverageNs"]) * 1e-9
for key, p in pmc.items():
    name = next((n for n in ("k_conv12_ws", "k_conv34_ws", "k_dense1_ws", "k_dense1_bf16x3", "k_conv12_bf16x3", "k_conv34_bf16x3") if n in key), None)
    if not name or "SQ_INSTS_MFMA" not in p:
        continue
    t = next((v for k, v in times.items() if name in k), None)
    if not t:
        continue
    clock = p["GRBM_GUI_ACTIVE"] / 8 / t if "GRBM_GUI_ACTIVE" in p else 1.9e9
    mfma, valu = p["SQ_INSTS_MFMA"], p["SQ_INSTS_VALU"] - p["SQ_INSTS_MFMA"]
    busy = p["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / clock
    waves = p["SQ_WAVE_CYCLES"]
    print(f"{name}: {t * 1e3:.3f} ms per 14 240 windows at {clock / 1e9:.2f} GHz (profiler)")
    print(f"   matrix cores busy {busy * 1e3:.3f} ms = {busy / t:.2f} of the kernel ({mfma / 1e6:.2f} M MFMAs, 32 cycles each per SIMD)")
    print(f"   per MFMA: {valu / mfma:.2f} VALU, {p['SQ_INSTS_SALU'] / mfma:.2f} scalar, {p['SQ_INSTS_LDS'] / mfma:.2f} LDS instructions "
          f"({p.get('SQ_INSTS_VALU_CVT', 0) / mfma:.2f} conversions per MFMA among the VALU)")
    lds = p["SQ_LDS_IDX_ACTIVE"] / CUS / clock
    print(f"   LDS array {lds * 1e3:.3f} ms = {lds / t:.2f} of the kernel, {p['SQ_LDS_BANK_CONFLICT'] / p['SQ_LDS_IDX_ACTIVE']:.2f} of it bank conflicts")
    print(f"   wave time: {100 * p['SQ_ACTIVE_INST_ANY'] / waves:.0f} % issuing, {100 * p['SQ_WAIT_INST_ANY'] / waves:.0f} % stalled at issue "
          f"(matrix pipe busy / operand not ready), {100 * p['SQ_WAIT_ANY'] / waves:.0f} % parked (s_waitcnt, barrier)")
