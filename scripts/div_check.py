"""bp_div_core / bp_div_prod / bp_ref_atanh vs IEEE and mpmath (diagnostics)."""
import os, sys, math, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpmath, torch, ldpc_amd
mpmath.mp.dps = 50
L = ldpc_amd.lib(); fn = L.ldpc_internal_bp_math; fn.restype = C.c_int; fn.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
def run(kind, x, n):
    d = torch.from_numpy(x).to("cuda"); o = torch.empty(n, dtype=torch.float64, device="cuda"); assert fn(kind, n, d.data_ptr(), o.data_ptr()) == 0; return o.cpu().numpy()
rng = np.random.default_rng(5)
n = 1 << 22
for name, a, b in (("unit", rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)),
                   ("wide", rng.uniform(-1, 1, n) * 2.0 ** rng.integers(-400, 400, n), rng.uniform(0.5, 1, n) * 2.0 ** rng.integers(-400, 400, n)),
                   ("log1p site", rng.uniform(-0.3, 0.42, n), None), ("atanh site", None, None)):
    if name == "log1p site": b = 2.0 + a
    if name == "atanh site":
        y = np.concatenate([rng.uniform(-1, 1, n // 2), 1 - 2.0 ** rng.uniform(-53, -1, n // 2)]); a = 2 * y; b = 1 - y
    got = run(4, np.concatenate([a, b]), n); ref = a / b
    bad = got != ref
    print("%-12s div_core != IEEE: %d of %d (%.2e)" % (name, bad.sum(), n, bad.mean()), "max ulps", np.max(np.abs(got[bad] - ref[bad]) / np.spacing(np.abs(ref[bad]))) if bad.any() else 0)
# guarded version on nasty operands
a = np.array([0.0, 0.0, 1e-320, 1e-300, 0.5, np.nan, 1e-310, -0.0]); b = np.array([0.0, 1.0, 1e-310, 1e-300, 0.0, 1.0, 1e-310, 0.0])
pad = 64 - len(a); a = np.concatenate([a, np.full(pad, 0.25)]); b = np.concatenate([b, np.full(pad, 0.5)])
got = run(6, np.concatenate([a, b]), 64)
with np.errstate(all="ignore"): ref = a / b
print("div_prod edge cases equal IEEE:", all((g == r) or (g != g and r != r) for g, r in zip(got, ref)), got[:8], ref[:8])
