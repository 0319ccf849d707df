#!/bin/bash
# Diagnostic: bench.py workloads on library variants, alternating on one box.
# usage: WORKLOAD="cfg3 --fft f64" tools/ab_variants_bench.sh <tag> <variant> [<variant> ...]   ("tree" = the tree's library)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
for v in "$@"; do
  L=tools/libf2cnn_hip_$v.so; [ $v = tree ] && L=f2cnn_amd/lib/libf2cnn_hip.so
  timeout -k 10 200 python tools/bench_with_lib.py $L --workload ${WORKLOAD:-cfg3} --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline $EXTRA > gpurun_out/ab_${v}.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_${v}.json")); k=d.get("kernels", {})
print("$v", d["value"], d["ms_per_step"], {n:round(x["ms_per_step"] if isinstance(x, dict) else x,3) for n,x in k.items()}, flush=True)
PY
done; done | tee gpurun_out/${tag}_ab_bench.txt
