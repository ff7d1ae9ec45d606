#!/bin/bash
# Pipe-utilisation counters of the flip kernel at C2 (one rocprofv3 --pmc pass, 8 SQ counters + GRBM).
#   bash scripts/flip_pmc2.sh gpurun_out/flip_pmc2 [shots]
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/flip_pmc2}")
SHOTS=${2:-4194304}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    -d "$OUT/pipes" -o pmc --output-format csv -- python3 "$ROOT/scripts/flip_prof.py" "$SHOTS" > "$OUT/pipes.log" 2>&1
echo done
