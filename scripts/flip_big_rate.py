"""Flip decoder at configs[4] (n = 10^6): rate of flip_big_kernel (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd
if os.environ.get("AB_LIB"):   # a variant library (scripts/build_variant.sh)
    ldpc_amd.LIB_PATH = os.environ["AB_LIB"]
from ldpc_amd import FlipDecoder, LinearCode
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
shots = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
code = LinearCode.random_regular_code(1000000, 500000, 3, 6, seed=1)
dec = FlipDecoder(code)
dec.mc_run(p, 1, 0, 64)
c = dec.mc_run(p, 2024, 0, shots)
ms = ldpc_amd.last_kernel_ms()
print("%s p=%g shots=%d %.1f ms -> %.3e shots/s, flips/shot %.0f, failures %d" % (ldpc_amd.last_kernel_name(), p, shots, ms, shots / ms * 1e3, int(c[4]) / shots, int(c[1])))
