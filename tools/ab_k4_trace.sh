#!/bin/bash
# Diagnostic: per-kernel times of the CNN (rocprofv3 --kernel-trace --stats of tools/k4_probe.py <n>) for library variants, alternating
# on one box. usage: tools/ab_k4_trace.sh <tag> <n> <variant> [<variant> ...]     ("tree" = the tree's library)
set -e
tag=$1; n=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  for v in "$@"; do
    L=tools/libf2cnn_hip_$v.so; [ $v = tree ] && L=f2cnn_amd/lib/libf2cnn_hip.so
    export F2CNN_PROBE_LIB=$L
    d=gpurun_out/${tag}_kt_${v}_$i
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/k4_probe.py $n > $d.log 2>&1
    python3 - $d $v <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    m = re.search(r"k_[a-z0-9_]+", r["Name"])
    if m and int(r["Calls"]) >= 10:
        out.append(f"{m.group(0)} {float(r['AverageNs']) / 1e3:.1f} us (min {float(r['MinNs']) / 1e3:.1f})")
print(f"[{sys.argv[2]}] " + "; ".join(out))
PY
  done
done | tee gpurun_out/${tag}_ab_trace.txt
