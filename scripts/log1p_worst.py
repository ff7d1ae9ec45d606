import os, sys, math, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ldpc_amd
L = ldpc_amd.lib(); fn = L.ldpc_internal_bp_math; fn.restype = C.c_int; fn.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
rng = np.random.default_rng(2)
xs = np.concatenate([rng.uniform(-0.999999, 5, 4000), 10.0 ** rng.uniform(-300, 300, 2000),
                     -(10.0 ** rng.uniform(-300, -1e-9, 1000)), rng.uniform(-0.30, -0.28, 300),
                     rng.uniform(0.40, 0.43, 300), -1 + 10.0 ** rng.uniform(-16, -1, 500),
                     np.array([0.41421356237309515, -0.2928932188134524, 2.0 ** 53, 2.0 ** 54, 1e-320])])
d = torch.from_numpy(xs).to("cuda"); o = torch.empty_like(d); fn(1, d.numel(), d.data_ptr(), o.data_ptr())
got = o.cpu().numpy(); ref = np.log1p(xs)
err = np.abs(got - ref) / np.spacing(np.abs(ref))
idx = np.argsort(-err)[:12]
for i in idx: print(repr(xs[i]), repr(got[i]), repr(ref[i]), err[i])
