#!/bin/bash
# Pipe-utilisation counters of the LDS-resident and of the HBM BP form at C1 (one rocprofv3 --pmc pass each).
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/bp_lds_pmc}")
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for path in lds hbm; do
  LDPC_HIP_BP_PATH=$path timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    -d "$OUT/$path" -o pmc --output-format csv -- python3 "$ROOT/scripts/bp_one.py" c1 4194304 0.02 > "$OUT/$path.log" 2>&1
done
echo done
