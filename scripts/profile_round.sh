#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*.  Run on the GPU box from the repo root:
#   bash scripts/profile_round.sh gpurun_out/prof
# One pass per counter group (never --pmc together with other trace domains); the program itself after `--`.
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/prof}")
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
FLIP="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bp"
BP="python3 $ROOT/scripts/bp_one.py c2 8388608 0.02"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/flip_stats" -o flip --output-format csv -- $FLIP > "$OUT/flip_stats.log" 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/bp_stats" -o bp --output-format csv -- $BP > "$OUT/bp_stats.log" 2>&1
pass() {  # name, program, counters...
    local name=$1 prog=$2; shift 2
    timeout -s KILL 300 rocprofv3 --pmc "$@" -d "$OUT/$name" -o pmc --output-format csv -- $prog > "$OUT/$name.log" 2>&1
}
pass flip_rd "$FLIP" TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass flip_wr "$FLIP" TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass flip_sq "$FLIP" SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass flip_lds "$FLIP" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY TCC_HIT_sum TCC_MISS_sum
pass bp_rd "$BP" TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass bp_wr "$BP" TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass bp_sq "$BP" SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
echo "profiles written to $OUT"
