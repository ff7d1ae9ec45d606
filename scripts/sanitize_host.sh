#!/bin/bash
# Host-side code under AddressSanitizer + UBSan (CPU only; the GPU code is not sanitized on this pool):
# the JSON reader on 300,000 mutated documents, the GF(2) algebra / code constructors on random inputs.
set -e -o pipefail
cd "$(dirname "$0")/../ldpc_amd/csrc"
OUT=${TMPDIR:-/tmp}
FLAGS="-std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -I."
g++ $FLAGS ../../scripts/fuzz_json.cpp host_json.cpp host_codes.cpp -o "$OUT/fuzz_json"
g++ $FLAGS ../../scripts/fuzz_host.cpp host_codes.cpp -o "$OUT/fuzz_host"
"$OUT/fuzz_json"
"$OUT/fuzz_host"
