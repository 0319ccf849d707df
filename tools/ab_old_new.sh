# Diagnostic: same-box A/B of two library builds (tools/libf2cnn_hip_old.so against the tree's) on bench workloads.
set -e
for i in 1 2 3; do
for lib in old new; do
  L=tools/libf2cnn_hip_$lib.so; [ $lib = new ] && L=f2cnn_amd/lib/libf2cnn_hip.so
  F2CNN_PROBE_OLD_LIB=1 timeout -k 10 200 python tools/bench_with_lib.py $L --workload ${WORKLOAD:-cfg3} --steps 20 --warmup 5 --no-cpu-baseline $EXTRA > gpurun_out/ab_${lib}.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_${lib}.json")); k=d["kernels"]
print("$lib", d["value"], d["ms_per_step"], {n:round(v["ms_per_step"],3) for n,v in k.items()}, flush=True)
PY
done; done
