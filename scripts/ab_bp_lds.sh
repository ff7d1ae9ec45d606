#!/bin/bash
# Same-box A/B: LDS-resident BP form against the HBM forms on C1 and the CSS example.
set -e -o pipefail
for rep in 1 2; do
  for path in hbm lds; do
    echo "== LDPC_HIP_BP_PATH=$path"
    LDPC_HIP_BP_PATH=$path timeout -k 10 200 python3 -u scripts/bp_one.py c1 4194304 0.02 | tail -1
    LDPC_HIP_BP_PATH=$path timeout -k 10 200 python3 -u scripts/css_one.py 4194304 | tail -1
  done
done
