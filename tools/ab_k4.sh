# Diagnostic: same-box alternation of CNN library variants (tools/libf2cnn_hip_<name>.so ...; "tree" = the tree's library)
# usage: tools/ab_k4.sh <tag> <variant> [<variant> ...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  L=tools/libf2cnn_hip_$v.so; [ $v = tree ] && L=f2cnn_amd/lib/libf2cnn_hip.so
  F2CNN_PROBE_LIB=$L timeout -k 10 300 python tests/diag/cnn_ws_check.py > gpurun_out/${tag}_ws_check_$v.txt 2>&1
  echo "[$v]"; tail -3 gpurun_out/${tag}_ws_check_$v.txt
done
for i in 1 2 3; do
  for v in "$@"; do
    L=tools/libf2cnn_hip_$v.so; [ $v = tree ] && L=f2cnn_amd/lib/libf2cnn_hip.so
    F2CNN_PROBE_LIB=$L timeout -k 10 120 python tools/k4_probe.py 14240 2>/dev/null | grep -i "conv\|cnn\|dense" | tr '\n' ' '; echo " [$v]"
  done
done | tee gpurun_out/${tag}_ab.txt
timeout -k 10 120 python tools/ws_stamps.py 14240 2>&1 | grep -v amdgpu.ids | cut -c1-200 > gpurun_out/${tag}_ws_stamps.txt
head -9 gpurun_out/${tag}_ws_stamps.txt
