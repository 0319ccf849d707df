#!/usr/bin/env python3
"""profiles/rNN_x_traffic_cfg3.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the cfg3 bench.
usage: make_traffic_json.py <fetch dir> <write dir> <out.json> [batch]"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def means(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = means(sys.argv[1], "FETCH_SIZE")
write, _ = means(sys.argv[2], "WRITE_SIZE")
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
C, N = 128, 16000
out = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc.sh) on `bench.py --workload cfg3 "
                 "--steps 2 --warmup 1 --no-cpu-baseline`; counters are KiB per dispatch, mean over the dispatches seen. "
                 "gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies the 128-byte requests of wide streaming reads "
                 "at 64 bytes, so it is doubled; WRITE_SIZE is exact. The reads of k_spectral_envelope are its channel tables and "
                 "utterance spectra missing the XCD's L2 (served by the Infinity Cache, which these counters include): cache "
                 "traffic of 17 MB of tables + 64 MB of spectra, not algorithmic bytes.",
       "workload": "cfg3", "batch": B, "lib_source_hash": bench.lib_source_hash()}
names = {"k_erb_filterbank": "k_erb_filterbank", "k_envelope<float, 13>": "k_envelope",
         "k_spectral_envelope<13": "k_spectral_envelope", "k_utterance_spectrum": "k_utterance_spectrum", "k_tail_state": "k_tail_state"}
# bytes each kernel MUST move: the spectral kernel writes the envelopes, its pre-kernels read the waves once each;
# when the spectral route serves the whole batch the two-kernel launches only skip utterances (nothing required of them)
spectral = any("k_spectral_envelope" in k for k in fetch)
need = {"k_erb_filterbank": 0 if spectral else B * 2 * N, "k_envelope": 0 if spectral else B * 8 * C * N,
        "k_spectral_envelope": B * 8 * C * N, "k_utterance_spectrum": B * 2 * N, "k_tail_state": 0}
for pat, key in names.items():
    kf = [k for k in fetch if pat in k]
    kw = [k for k in write if pat in k]
    if not kf or not kw:
        continue
    f, w = fetch[kf[0]], write[kw[0]]
    out[key] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "dispatches": nf[kf[0]],
                "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024), "required_bytes_per_launch": need[key]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
