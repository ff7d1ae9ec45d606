#!/bin/bash
# rocprofv3 passes over one program: kernel stats + four counter passes (one group per run, never --pmc with a trace).
#   bash scripts/pmc_run.sh OUTDIR python3 scripts/bp_one.py c5 17664 0.02
# (run from the repo root on the GPU box; OUTDIR usually under gpurun_out/; the program itself after the directory)
set -e -o pipefail
OUT=$(realpath -m "$1"); shift
ROOT=$(pwd)
mkdir -p "$OUT"
PROG=()
for a in "$@"; do if [ -e "$ROOT/$a" ]; then PROG+=("$ROOT/$a"); else PROG+=("$a"); fi; done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- "${PROG[@]}" > "$OUT/stats.log" 2>&1
pass() {
    local name=$1; shift
    timeout -s KILL 400 rocprofv3 --pmc "$@" -d "$OUT/$name" -o pmc --output-format csv -- "${PROG[@]}" > "$OUT/$name.log" 2>&1
}
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass wait SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum
pass rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
grep -h "kernel_ms\|shots" "$OUT/stats.log" | tail -3
echo done
