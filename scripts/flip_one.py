"""One flip Monte-Carlo launch at C2 (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd
from ldpc_amd import FlipDecoder, LinearCode
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 163840
code = LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
dec = FlipDecoder(code)
dec.mc_run(0.02, 1, 0, 4096)
c = dec.mc_run(0.02, 2024, 0, shots)
print("kernel_ms", ldpc_amd.last_kernel_ms(), "shots", shots, "flips", int(c[4]))
