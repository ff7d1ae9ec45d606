#!/bin/bash
# rocprofv3 counter passes over one flip Monte-Carlo run at C2 (scripts/flip_prof.py).  Usage on the GPU box:
#   bash scripts/flip_pmc.sh gpurun_out/flip_pmc [shots]
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/flip_pmc}")
SHOTS=${2:-1048576}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PROG="python3 $ROOT/scripts/flip_prof.py $SHOTS"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o flip --output-format csv -- $PROG > "$OUT/stats.log" 2>&1
pass() {
    local name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$OUT/$name" -o pmc --output-format csv -- $PROG > "$OUT/$name.log" 2>&1
}
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum
pass rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
echo done
