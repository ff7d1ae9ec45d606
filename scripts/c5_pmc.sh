#!/bin/bash
# diagnostics: counters of the streamed BP kernels at C5, phases not overlapped
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/c5pmc}"); ROOT=$(pwd); mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

P="python3 $ROOT/scripts/bp_one.py c5 4416 0.02"
timeout -s KILL 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/sq" -o pmc --output-format csv -- $P > "$OUT/sq.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum -d "$OUT/tcp" -o pmc --output-format csv -- $P > "$OUT/tcp.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum -d "$OUT/tcc" -o pmc --output-format csv -- $P > "$OUT/tcc.log" 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace -d "$OUT/trace" -o t --output-format csv -- $P > "$OUT/trace.log" 2>&1
echo done
