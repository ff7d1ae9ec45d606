import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ldpc_amd
L = ldpc_amd.lib(); fn = L.ldpc_internal_bp_math; fn.restype = C.c_int; fn.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
xs = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 100.0, -100.0, 1e300, -1e300, 64.0, 63.0, 40.0])
d = torch.from_numpy(xs).to("cuda"); o = torch.empty_like(d); fn(0, d.numel(), d.data_ptr(), o.data_ptr())
for x, g in zip(xs, o.cpu().numpy()): print(repr(x), repr(float(g)), float(g).hex())
