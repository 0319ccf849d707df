#!/bin/bash
# Diagnostic: per-kernel times inside the cfg4 bench step (rocprofv3 --kernel-trace --stats of tools/bench_with_lib.py) for library variants.
# usage: tools/ab_cfg4_trace.sh <tag> <variant> [<variant> ...]     ("tree" = the tree's library)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  L=tools/libf2cnn_hip_$v.so; [ $v = tree ] && L=f2cnn_amd/lib/libf2cnn_hip.so
  d=gpurun_out/${tag}_c4_${v}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_with_lib.py $L --workload cfg4 --steps 4 --warmup 1 --no-cpu-baseline > $d.json 2> $d.err || { echo "[$v] failed"; tail -3 $d.err; continue; }
  python3 - $d $v <<'PY'
import csv, glob, json, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    m = re.search(r"k_[a-z0-9_]+", r["Name"])
    if m and m.group(0) in ("k_conv12_ws", "k_conv34_ws", "k_dense1_ws", "k_eval_windows"):
        out.append(f"{m.group(0)} {float(r['AverageNs']) / 1e3:.1f} us x {r['Calls']} (max {float(r['MaxNs']) / 1e3:.1f})")
d = json.load(open(sys.argv[1] + ".json"))
print(f"[{sys.argv[2]}] {d['value']} audio-s/s, {d['ms_per_step']} ms; " + "; ".join(out))
PY
done | tee gpurun_out/${tag}_cfg4_trace.txt
