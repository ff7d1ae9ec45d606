#!/bin/bash
# diagnostics: counters of the persistent BP kernel on the CSS workload (C3)
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/csspmc}"); ROOT=$(pwd); mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

P="python3 $ROOT/scripts/css_one.py 262144"
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/sq" -o pmc --output-format csv -- $P > "$OUT/sq.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_LDS SQ_WAIT_ANY -d "$OUT/sq2" -o pmc --output-format csv -- $P > "$OUT/sq2.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_sum -d "$OUT/tcc" -o pmc --output-format csv -- $P > "$OUT/tcc.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d "$OUT/tcc2" -o pmc --output-format csv -- $P > "$OUT/tcc2.log" 2>&1
echo done
