#!/bin/bash
# Same-box A/B of two builds of the library (ab/libldpc_hip_base.so against the in-tree one): BP on C1, C2, the CSS example.
set -e -o pipefail
for rep in 1 2 3; do
  for lib in ab/libldpc_hip_base.so ldpc_amd/libldpc_hip.so; do
    echo "== $lib"
    AB_LIB=$lib timeout -k 10 200 python3 -u scripts/bp_one.py c1 4194304 0.02 | tail -1
    AB_LIB=$lib timeout -k 10 200 python3 -u scripts/css_one.py 4194304 | tail -1
    if [ "$rep" != 3 ]; then AB_LIB=$lib timeout -k 10 200 python3 -u scripts/bp_one.py c2 1048576 0.02 | tail -1; fi
  done
done
