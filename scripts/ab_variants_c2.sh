#!/bin/bash
# Same-box comparison of several builds of the library (ab/lib_*.so): BP on C2 (persistent kernel), C5 (streamed form), the CSS example on the HBM form.
set -e -o pipefail
for rep in 1 2; do
  for lib in ab/lib_*.so; do
    echo "== $lib"
    AB_LIB=$lib timeout -k 10 200 python3 -u scripts/bp_one.py c2 2097152 0.02 | tail -1
    AB_LIB=$lib timeout -k 10 300 python3 -u scripts/bp_one.py c5 17664 0.02 | tail -1
    AB_LIB=$lib LDPC_HIP_BP_PATH=hbm timeout -k 10 200 python3 -u scripts/css_one.py 1048576 | tail -1
  done
done
