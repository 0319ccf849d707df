set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -m pytest tests/test_gpu_windows_cnn.py -x -q > gpurun_out/r05_j_cnn_tests.log 2>&1; tail -2 gpurun_out/r05_j_cnn_tests.log
for i in 1 2 3; do
for lib in pre tree; do
  L=tools/libf2cnn_hip_$lib.so; [ $lib = tree ] && L=f2cnn_amd/lib/libf2cnn_hip.so
  timeout -k 10 200 python tools/bench_with_lib.py $L --workload cfg4 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ab_${lib}.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_${lib}.json"))
print("$lib", d["value"], d["ms_per_step"], flush=True)
PY
done; done | tee gpurun_out/r05_j_ab_cfg4.txt
tools/ab_k4_trace.sh r05_j 14240 pre tree
