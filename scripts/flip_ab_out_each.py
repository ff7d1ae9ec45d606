"""Flip Monte-Carlo at C2: what each recorded output costs (A/B aid; AB_LIB picks a variant library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldpc_amd
if os.environ.get("AB_LIB"):
    ldpc_amd.LIB_PATH = os.environ["AB_LIB"]
from ldpc_amd import FlipDecoder, LinearCode, nwords
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
code = LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
dec = FlipDecoder(code)
e = torch.zeros((shots, nwords(10000)), dtype=torch.int64, device="cuda")
s = torch.zeros((shots, nwords(5000)), dtype=torch.int64, device="cuda")
o = torch.zeros((shots, nwords(10000)), dtype=torch.int64, device="cuda")
dec.mc_run(0.02, 1, 0, 1 << 18, e, s, o)
for name, args in (("none", (None, None, None)), ("errors", (e, None, None)), ("syndromes", (None, s, None)), ("words", (None, None, o)),
                   ("all", (e, s, o)), ("none", (None, None, None)), ("all", (e, s, o))):
    best = 1e9
    for rep in range(3):
        c = dec.mc_run(0.02, 2024, 0, shots, *args)
        best = min(best, ldpc_amd.last_kernel_ms())
    print("%-10s kernel_ms %.3f -> %.3e shots/s" % (name, best, shots / best * 1e3), c[:5].tolist(), flush=True)
