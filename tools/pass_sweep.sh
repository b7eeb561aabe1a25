#!/bin/bash
# sweeps the k_pass geometry on the GPU box: ./tools/pass_sweep.sh n d k
n=${1:-100000}; d=${2:-10000}; k=${3:-50}
for u in 4 8 16; do
  for nt in 0 1; do
    for wgs in 2048 4096 8192; do
      RRI_PASS_UNROLL=$u RRI_PASS_NT=$nt RRI_PASS_WGS=$wgs timeout -k 5 120 python tools/pass_probe.py $n $d $k 2>&1 | grep -v amdgpu.ids
    done
  done
done
