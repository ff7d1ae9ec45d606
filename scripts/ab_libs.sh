#!/bin/bash
# A/B aid (GPU box): flip Monte-Carlo at C2 with each of the given variant libraries (built by scripts/build_variant.sh).
#   bash scripts/ab_libs.sh <out file> <shots> <LDPC_HIP_FLIP_NF selection> lib1.so lib2.so ...
out=$1; shots=$2; sel=$3; shift 3
: > "$out"
for lib in "$@"; do
  echo "== $lib" >> "$out"
  AB_LIB=$(realpath "$lib") timeout -k 10 200 python scripts/flip_ab.py "$shots" $sel >> "$out" 2>&1 || exit 1
done
grep -v amdgpu.ids "$out"
