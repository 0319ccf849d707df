#!/usr/bin/env python3
"""profiles/rNN_x_traffic_cfg5r.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py --workload cfg5r --corpus N --steps 1 --warmup 1 --no-cpu-baseline`: HBM-side bytes per launch of every kernel of the
ragged pass, and per ROW for the kernels that own rows (gfx950: FETCH_SIZE doubled, MI355X_MICROARCH.md HBM section).
usage: make_traffic_cfg5r.py <fetch dir> <write dir> <out.json> <corpus>"""
import collections, csv, glob, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def means(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
corpus = int(sys.argv[4])
lens = bench.ragged_lengths(2029, corpus)
C = 128
launches = max(1, -(-corpus // 2500))
cls = {"k_spectral_envelope<13": (lens > 8192) & (lens <= 16384 - 256), "k_spectral_envelope<14": (lens > 16384) & (lens <= 32768 - 256),
       "k_spectral_envelope_long": (lens > 32768) & (lens <= 65536 - 256)}
out = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --workload cfg5r; KiB per dispatch, mean over "
                 "dispatches; hbm bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction); rows = utterances of the class x 128 channels per launch",
       "workload": "cfg5r", "corpus": corpus, "launches_per_pass": launches, "lib_source_hash": bench.lib_source_hash()}
for k in sorted(fetch):
    if k not in write:
        continue
    f, nf = fetch[k]
    w, _ = write[k]
    e = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "dispatches": nf, "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024)}
    hit = next((c for c in cls if c in k), None)
    if hit:
        sel = cls[hit]
        rows = int(sel.sum()) * C / launches
        samples = float(lens[sel].sum()) * C / launches
        e["rows_per_launch"] = round(rows, 1)
        e["hbm_bytes_per_row"] = int(e["hbm_bytes_per_launch"] / max(rows, 1))
        e["envelope_bytes_per_row"] = int(8 * samples / max(rows, 1))
    out[k[:70]] = e
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
