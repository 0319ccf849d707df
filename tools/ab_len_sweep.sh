#!/bin/bash
# Diagnostic: tools/len_sweep.py on library variants, alternating on one box. usage: tools/ab_len_sweep.sh <tag> "<total> <len> ..." <variant> ...
tag=$1; args=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  for v in "$@"; do
    L=tools/libf2cnn_hip_$v.so; [ $v = tree ] && L=f2cnn_amd/lib/libf2cnn_hip.so
    echo "[$v] $(F2CNN_PROBE_LIB=$L timeout -k 10 200 python tools/len_sweep.py $args 2>/dev/null | grep '=>' | sed 's/B=.*=>//; s/audio-s.s (kernel time only)//' | tr '\n' ' ')"
  done
done | tee gpurun_out/${tag}_ab_len_sweep.txt
