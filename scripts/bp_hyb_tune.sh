#!/bin/bash
# Hybrid LDS form on C2: workgroup width and LDS share per workgroup (diagnostics).
set -e -o pipefail
run() { echo "== threads=${LDPC_HIP_BP_LDS_THREADS:-auto} kb=${LDPC_HIP_BP_HYB_KB:-150}"; timeout -k 10 200 python3 -u scripts/bp_one.py c2 1048576 0.02 | tail -1; }
LDPC_HIP_BP_PATH=hbm run
run
LDPC_HIP_BP_LDS_THREADS=768 run
LDPC_HIP_BP_LDS_THREADS=512 run
LDPC_HIP_BP_HYB_KB=158 run
LDPC_HIP_BP_HYB_KB=120 run
LDPC_HIP_BP_HYB_KB=78 LDPC_HIP_BP_LDS_THREADS=768 run
LDPC_HIP_BP_HYB_KB=78 LDPC_HIP_BP_LDS_THREADS=512 run
LDPC_HIP_BP_HYB_KB=52 LDPC_HIP_BP_LDS_THREADS=512 run
LDPC_HIP_BP_HYB_KB=38 LDPC_HIP_BP_LDS_THREADS=384 run
