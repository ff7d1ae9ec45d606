"""Do the BP runs that never converge (CSS example, degenerate code) reach a state that repeats?  Compares the
check -> bit messages of such shots after k and k + 1 (and k + 2) iterations, bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ldpc_amd
from ldpc_amd import BpDecoder, CssCode, LinearCode, DepolarizingNoise, nwords
base = LinearCode.random_regular_code(20, 15, 3, 4, seed=1)
css = CssCode.hypergraph_product(base, base)
hx = css.x_stabs_binary()          # decodes the z part of the error
n, m, E = hx.len(), hx.num_checks(), hx.num_edges()
p, shots = 0.01, 4096
dx = torch.zeros((shots, nwords(n)), dtype=torch.int64, device="cuda"); dz = torch.zeros_like(dx)
DepolarizingNoise(p).sample_batch(n, 1, 0, shots, dx, dz)
syn = torch.zeros((shots, nwords(m)), dtype=torch.int64, device="cuda")
hx.syndrome_batch(shots, dz, syn)
def run(cap, sel=None):
    s = syn if sel is None else syn[sel].contiguous()
    k = s.shape[0]
    dec = BpDecoder(hx, 2 * p / 3, cap)
    c = torch.zeros((k, nwords(n)), dtype=torch.int64, device="cuda"); it = torch.zeros((k,), dtype=torch.int32, device="cuda")
    post = torch.zeros((k, n), dtype=torch.float64, device="cuda"); msg = torch.zeros((k, E), dtype=torch.float64, device="cuda")
    dec.correction_for_batch(k, s, c, it, post, msg)
    return c.cpu().numpy(), it.cpu().numpy(), msg.cpu().numpy().view(np.uint64)
_, it100, _ = run(100)
bad = np.nonzero(it100 == 100)[0]
print("shots", shots, "never converge:", len(bad))
sel = torch.tensor(bad, device="cuda")
prev = None
for cap in (5, 6, 7, 10, 11, 12, 20, 21, 22, 40, 41, 42, 80, 81, 82, 99, 100):
    c, it, msg = run(cap, sel)
    if prev is not None and prev[0] == cap - 1:
        same1 = (msg == prev[1]).all(axis=1).mean()
        print("cap %3d: messages identical to cap %3d on %.3f of the shots" % (cap, cap - 1, same1), end="")
        if prev2 is not None and prev2[0] == cap - 2:
            print("; to cap %3d on %.3f" % (cap - 2, (msg == prev2[1]).all(axis=1).mean()), end="")
        print()
    prev2 = prev
    prev = (cap, msg)
