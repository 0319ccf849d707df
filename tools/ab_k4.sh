set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python tests/diag/cnn_ws_check.py > gpurun_out/r04_c_ws_check.txt 2>&1
tail -5 gpurun_out/r04_c_ws_check.txt
for i in 1 2 3; do
  F2CNN_PROBE_LIB=tools/libf2cnn_hip_old.so timeout -k 10 120 python tools/k4_probe.py 14240 2>/dev/null | grep -i "conv\|cnn" | tr '\n' ' '; echo " [old]"
  timeout -k 10 120 python tools/k4_probe.py 14240 2>/dev/null | grep -i "conv\|cnn" | tr '\n' ' '; echo " [new]"
done
timeout -k 10 120 python tools/ws_stamps.py 14240 2>&1 | grep -v amdgpu.ids | cut -c1-200 > gpurun_out/r04_c_ws_stamps.txt
head -9 gpurun_out/r04_c_ws_stamps.txt
