#!/bin/bash
# Same-box A/B of two builds of the library on the BP workloads: ab/libldpc_hip_base.so against the tree's.
#   bash scripts/ab_bp.sh > gpurun_out/ab_bp.log
set -e -o pipefail
for rep in 1 2; do
  for lib in ab/libldpc_hip_base.so ldpc_amd/libldpc_hip.so; do
    echo "== $lib"
    AB_LIB=$PWD/$lib timeout -k 10 200 python3 -u scripts/bp_one.py c2 4194304 0.02 | tail -1
    AB_LIB=$PWD/$lib timeout -k 10 200 python3 -u scripts/bp_one.py c1 4194304 0.02 | tail -1
    AB_LIB=$PWD/$lib timeout -k 10 200 python3 -u scripts/css_one.py 4194304 | tail -1
  done
done
