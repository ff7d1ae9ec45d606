#!/bin/bash
# Same-box comparison of several builds of the library (ab/lib_*.so): BP on C2.
set -e -o pipefail
for rep in 1 2; do
  for lib in ab/lib_*.so; do
    echo "== $lib"
    AB_LIB=$lib timeout -k 10 200 python3 -u scripts/bp_one.py c2 1048576 0.02 | tail -1
  done
done
