#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>  ->  tools/libf2cnn_hip_<name>.so
# Diagnostic variants of the library (phase stamps, knock-outs, alternative plans). Never used for results.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p /tmp/f2var_$name
for f in f2cnn_amd/csrc/*.hip; do
  extra=""; case "$f" in *envelope*|*spectral*) extra="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" $extra -c "$f" -o /tmp/f2var_$name/$(basename "$f" .hip).o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/f2var_$name/*.o -o tools/libf2cnn_hip_$name.so
