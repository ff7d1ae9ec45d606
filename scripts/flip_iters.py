import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd
from ldpc_amd import FlipDecoder, LinearCode
code = LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
dec = FlipDecoder(code)
c = dec.mc_run(0.02, 1, 0, 4096)
print("lanes", os.environ.get("LDPC_HIP_FLIP_LANES"), "flips/shot", c[4]/4096, "iterations/shot", c[5]/4096)
