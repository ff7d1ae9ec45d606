#!/bin/bash
# Same-box A/B of variant libraries on the BP workloads (C2 hybrid, C1, CSS example): bash scripts/ab_bp_all.sh <out> lib1.so lib2.so ...
out=$1; shift
: > "$out"
for rep in 1 2; do
for lib in "$@"; do
  echo "== $lib" >> "$out"
  AB_LIB=$(realpath "$lib") timeout -k 10 300 python3 -u scripts/bp_one.py c2 4194304 0.02 2>&1 | tail -1 >> "$out" || exit 1
  AB_LIB=$(realpath "$lib") timeout -k 10 300 python3 -u scripts/bp_one.py c1 4194304 0.02 2>&1 | tail -1 >> "$out" || exit 1
  AB_LIB=$(realpath "$lib") timeout -k 10 300 python3 -u scripts/css_one.py 4194304 2>&1 | tail -1 >> "$out" || exit 1
done
done
cat "$out"
