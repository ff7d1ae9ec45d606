"""A/B of the flip4 forms at C2 (LDPC_HIP_FLIP_NF values on the command line); counters must agree."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd
if os.environ.get("AB_LIB"):   # A/B aid: another build of the library
    ldpc_amd.LIB_PATH = os.environ["AB_LIB"]
from ldpc_amd import FlipDecoder, LinearCode
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
sel = sys.argv[2:] or ["9"]
code = LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
dec = FlipDecoder(code)
ref = None
for s in sel:
    os.environ["LDPC_HIP_FLIP_NF"] = s
    ldpc_amd.reload_switches()
    dec.mc_run(0.02, 1, 0, 1 << 18)
    best = 1e9
    for rep in range(3):
        c = dec.mc_run(0.02, 2024, 0, shots)
        best = min(best, ldpc_amd.last_kernel_ms())
    if ref is None:
        ref = c[:5].tolist()
    print("NF=%s %s: kernel_ms %.3f -> %.3e shots/s; counters %s %s" % (s, ldpc_amd.last_kernel_name(), best, shots / best * 1e3,
          c[:5].tolist(), "same" if c[:5].tolist() == ref else "DIFFER"), flush=True)
