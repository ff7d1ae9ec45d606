"""One BP Monte-Carlo launch (for rocprofv3 passes).  usage: bp_one.py [config] [shots] [p]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd
if os.environ.get("AB_LIB"):   # A/B aid: another build of the library
    ldpc_amd.LIB_PATH = os.environ["AB_LIB"]
from ldpc_amd import BpDecoder, LinearCode
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
shots = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
shape = {"c1": (1200, 900, 3, 4), "c2": (10000, 5000, 3, 6), "c5": (1000000, 500000, 3, 6)}[cfg]
code = LinearCode.random_regular_code(*shape, seed=1)
dec = BpDecoder(code, p, 100)
dec.mc_run(p, 1, 0, 4096 if cfg != "c5" else 64)
c = dec.mc_run(p, 2024, 0, shots)
ms = ldpc_amd.last_kernel_ms()
E = code.num_edges()
ran = ldpc_amd.last_bp_executed_iterations()   # (the LDS form ends periodic runs early: price what was run)
print("cfg", cfg, "kernel_ms", ms, "shots", shots, "iters", int(c[4]), "fail", int(c[1]), "ran", ran,
      "algorithmic_GBs", ran * 32 * E / (ms * 1e-3) / 1e9)
