# A/B of flip4 forms at C2 plus the phase stamps of the default form (diagnostics; LDPC_HIP_FLIP_NF / _PROF)
set -e
timeout -k 10 300 python scripts/flip_ab.py 4194304 ${FLIP_AB_SEL:-108 8 9 8} > gpurun_out/ab_e3.txt 2>&1
for nf in ${FLIP_PROF_SEL:-8}; do LDPC_HIP_FLIP_NF=$nf LDPC_HIP_FLIP_PROF=1 timeout -k 10 120 python scripts/flip_prof.py 4194304 >> gpurun_out/ab_e3.txt 2>&1; done
cat gpurun_out/ab_e3.txt
