"""LDS form against the HBM forms on codes of which one shot fits a CU's LDS, large calls (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd
from ldpc_amd import BpDecoder, LinearCode
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
for (n, m, dv, dc, p) in [(5000, 2500, 3, 6, 0.02), (6000, 3000, 3, 6, 0.02), (6000, 3000, 3, 6, 0.05), (4200, 1800, 3, 7, 0.01)]:
    code = LinearCode.random_regular_code(n, m, dv, dc, seed=1)
    ref = None
    for path in ("hbm", "lds", "default"):
        if path == "default": os.environ.pop("LDPC_HIP_BP_PATH", None)
        else: os.environ["LDPC_HIP_BP_PATH"] = path
        ldpc_amd.reload_switches()
        dec = BpDecoder(code, p, 100)
        dec.mc_run(p, 1, 0, 4096)
        c = dec.mc_run(p, 2024, 0, shots)
        ms = ldpc_amd.last_kernel_ms()
        if ref is None: ref = c.tolist()
        print("n=%d E=%d p=%.2f %-8s %9.2f ms  %.3e shots/s  %s  %s" % (n, code.num_edges(), p, path, ms, shots / ms * 1e3, ldpc_amd.last_kernel_name(),
              "same" if c.tolist() == ref else "DIFFER"), flush=True)
