#!/bin/bash
# Same-box comparison of several builds of the library (ab/lib_*.so): BP on C1 and the CSS example.
set -e -o pipefail
for rep in 1 2; do
  for lib in ab/lib_*.so; do
    echo "== $lib"
    AB_LIB=$lib timeout -k 10 200 python3 -u scripts/bp_one.py c1 4194304 0.02 | tail -1
    AB_LIB=$lib timeout -k 10 200 python3 -u scripts/css_one.py 4194304 | tail -1
  done
done
