#!/bin/bash
# Same-box A/B of variant libraries on the C2 BP workload (hybrid LDS form): bash scripts/ab_bp_c2.sh <out> <shots> lib1.so lib2.so ...
out=$1; shots=$2; shift 2
: > "$out"
for lib in "$@"; do
  echo "== $lib" >> "$out"
  AB_LIB=$(realpath "$lib") timeout -k 10 300 python3 -u scripts/bp_one.py c2 "$shots" 0.02 2>&1 | tail -1 >> "$out" || exit 1
done
cat "$out"
