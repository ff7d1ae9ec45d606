#!/bin/bash
# Same-box A/B of two builds of the library (ab/libldpc_hip_base.so against the in-tree one) on C1 and the CSS example.
set -e -o pipefail
for rep in 1 2 3; do
  for lib in ab/libldpc_hip_base.so ldpc_amd/libldpc_hip.so; do
    echo "== $lib"
    AB_LIB=$lib timeout -k 10 200 python3 -u scripts/bp_one.py c1 4194304 0.02 | tail -1
    AB_LIB=$lib timeout -k 10 200 python3 -u scripts/css_one.py 4194304 | tail -1
  done
done
