"""bp_lds_kernel's workgroup size on codes of which few shots fit a CU's LDS (diagnostics): LDPC_HIP_BP_LDS_THREADS values
against the choice of bp_lds_shape and against the HBM forms.  usage: bp_lds_threads.py [shots]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd
from ldpc_amd import BpDecoder, LinearCode
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
for (n, m, dv, dc, p) in [(1200, 900, 3, 4, 0.02), (2000, 1000, 3, 6, 0.02), (3000, 1500, 3, 6, 0.02), (5000, 2500, 3, 6, 0.02), (2100, 900, 3, 7, 0.01)]:
    code = LinearCode.random_regular_code(n, m, dv, dc, seed=1)
    ref = None
    for path, threads in [("hbm", None), ("lds", None), ("lds", 256), ("lds", 320), ("lds", 512), ("lds", 768), ("lds", 1024)]:
        os.environ["LDPC_HIP_BP_PATH"] = path
        if threads: os.environ["LDPC_HIP_BP_LDS_THREADS"] = str(threads)
        else: os.environ.pop("LDPC_HIP_BP_LDS_THREADS", None)
        ldpc_amd.reload_switches()
        dec = BpDecoder(code, p, 100)
        dec.mc_run(p, 1, 0, 4096)
        best = 1e9
        for rep in range(2):
            c = dec.mc_run(p, 2024, 0, shots)
            best = min(best, ldpc_amd.last_kernel_ms())
        if ref is None: ref = c.tolist()
        print("n=%d E=%d %-4s threads=%-6s %8.2f ms  %.3e shots/s  %s  %s" % (n, code.num_edges(), path, threads or "auto", best, shots / best * 1e3,
              ldpc_amd.last_kernel_name(), "same" if c.tolist() == ref else "DIFFER"), flush=True)
