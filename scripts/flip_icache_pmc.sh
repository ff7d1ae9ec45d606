#!/bin/bash
# Diagnostics (GPU box): instruction-cache counters of the flip kernel for each of the given variant libraries.
#   bash scripts/flip_icache_pmc.sh <out dir under gpurun_out> lib1.so lib2.so ...
set -e -o pipefail
OUT=$(realpath -m "$1"); shift
ROOT=$(pwd)
mkdir -p "$OUT"
LIBS=()
for l in "$@"; do LIBS+=("$(realpath "$l")"); done
cd /tmp && export TMPDIR=/tmp
for lib in "${LIBS[@]}"; do
  name=$(basename "$lib" .so)
  export AB_LIB="$lib"
  timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d "$OUT/$name/pass" -o pmc --output-format csv -- python3 "$ROOT/scripts/flip_prof.py" 4194304 > "$OUT/$name.log" 2>&1
  python3 "$ROOT/scripts/pmc_kernel.py" "$OUT/$name" flip4_kernel >> "$OUT/summary.txt" 2>&1 || true
done
cat "$OUT/summary.txt"
