#!/bin/bash
# A/B aid (GPU box): the flip kernel's phase stamps (LDPC_HIP_FLIP_PROF=1) for each of the given variant libraries.
#   bash scripts/prof_libs.sh <out file> <shots> lib1.so lib2.so ...
out=$1; shots=$2; shift 2
: > "$out"
for lib in "$@"; do
  echo "== $lib" >> "$out"
  AB_LIB=$(realpath "$lib") LDPC_HIP_FLIP_PROF=1 timeout -k 10 200 python scripts/flip_prof.py "$shots" >> "$out" 2>&1 || exit 1
done
grep -v amdgpu.ids "$out"
