#!/bin/bash
# usage (on the GPU box): tools/ks_variants.sh OUT name1 name2 ...  -- bench cfg3 with the tree's library and the named
# tools/libf2cnn_hip_<name>.so variants, alternating, and print the k_spectral_envelope / step times
out=$1; shift
mkdir -p $(dirname $out)
for round in 1 2; do
for v in tree "$@"; do
  if [ $v = tree ]; then
    python bench.py --workload cfg3 --steps 20 --warmup 3 > /tmp/ksv.json 2>/dev/null
  else
    python tools/bench_with_lib.py tools/libf2cnn_hip_$v.so --workload cfg3 --steps 20 --warmup 3 > /tmp/ksv.json 2>/dev/null
  fi
  python -c "
import json; d=json.load(open('/tmp/ksv.json')); k=d['kernels']
print('$v', 'step', d['ms_per_step'], {n:v['ms_per_step'] for n,v in k.items() if v['ms_per_step']>0.05})" | tee -a $out
done
done
