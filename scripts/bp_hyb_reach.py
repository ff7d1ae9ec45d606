"""Hybrid LDS form against bp_kernel on (3,6)-regular codes beyond C2 (diagnostics): where the L2-resident share stops paying."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd
from ldpc_amd import BpDecoder, LinearCode
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
for n in (8000, 10000, 12000, 14000, 16000, 20000):
    code = LinearCode.random_regular_code(n, n // 2, 3, 6, seed=1)
    ref = None
    for path in ("hbm", "lds"):
        os.environ["LDPC_HIP_BP_PATH"] = path
        ldpc_amd.reload_switches()
        dec = BpDecoder(code, 0.02, 100)
        dec.mc_run(0.02, 1, 0, 4096)
        c = dec.mc_run(0.02, 2024, 0, shots)
        ms = ldpc_amd.last_kernel_ms()
        if ref is None: ref = c.tolist()
        print("n=%d E=%d %-4s %9.2f ms  %.3e shots/s  %s  %s" % (n, code.num_edges(), path, ms, shots / ms * 1e3, ldpc_amd.last_kernel_name()[:24],
              "same" if c.tolist() == ref else "DIFFER"), flush=True)
