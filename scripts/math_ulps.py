"""ulp error statistics of bp_math.hpp vs OCML vs host libm (diagnostics)."""
import os, sys, math, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpmath, torch, ldpc_amd
mpmath.mp.dps = 50
L = ldpc_amd.lib(); fn = L.ldpc_internal_bp_math; fn.restype = C.c_int; fn.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
def run(kind, x):
    d = torch.from_numpy(x).to("cuda"); o = torch.empty_like(d); assert fn(kind, d.numel(), d.data_ptr(), o.data_ptr()) == 0; return o.cpu().numpy()
def stats(name, got, ref):
    err = []
    for g, r in zip(got, ref):
        rf = float(r)
        if rf == 0 or not math.isfinite(rf): continue
        err.append(abs(float((mpmath.mpf(float(g)) - r) / math.ulp(rf))))
    err = np.array(err)
    print("%-28s max %.3f ulp  mean %.3f  frac correctly rounded %.3f" % (name, err.max(), err.mean(), (err <= 0.5).mean()))
rng = np.random.default_rng(3)
for lo, hi in ((0.5, 2.0), (2.0, 8.0), (8.0, 40.0), (0.0, 0.5)):
    x = rng.uniform(lo, hi, 4000) * rng.choice([-1.0, 1.0], 4000)
    ref = [mpmath.tanh(mpmath.mpf(float(v)) / 2) for v in x]
    print("tanh(x/2), |x| in [%g, %g)" % (lo, hi))
    stats("  bp_tanh_half", run(0, x), ref); stats("  OCML tanh", run(2, x), ref); stats("  host libm tanh", np.tanh(x / 2), ref)
for lo, hi in ((-0.99, -0.3), (-0.29, 0.41), (0.42, 10.0), (10.0, 1e6)):
    x = rng.uniform(lo, hi, 4000)
    ref = [mpmath.log1p(mpmath.mpf(float(v))) for v in x]
    print("log1p(x), x in [%g, %g)" % (lo, hi))
    stats("  bp_log1p", run(1, x), ref); stats("  OCML log1p", run(3, x), ref); stats("  host libm log1p", np.log1p(x), ref)
