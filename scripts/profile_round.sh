#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*.  Run on the GPU box from the repo root:
#   bash scripts/profile_round.sh gpurun_out/prof
# then, here:  python scripts/summarize_profiles.py gpurun_out/prof r02
# One pass per counter group (scripts/pmc_run.sh: never --pmc together with a trace domain); the program itself after `--`.
set -e -o pipefail
OUT=${1:-gpurun_out/prof}
bash scripts/pmc_run.sh "$OUT/flip" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bp
bash scripts/pmc_run.sh "$OUT/bp" python3 scripts/bp_one.py c2 8388608 0.02
bash scripts/pmc_run.sh "$OUT/c5" python3 scripts/bp_one.py c5 70656 0.02
bash scripts/pmc_run.sh "$OUT/css" python3 scripts/css_one.py 8388608
bash scripts/pmc_run.sh "$OUT/c1" python3 scripts/bp_one.py c1 4194304 0.02
echo "profiles written to $OUT"
