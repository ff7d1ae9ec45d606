"""LDS-resident BP form against the HBM forms on small calls (wall time of mc_run, counters must agree)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd as L

def run(dec, p, shots, path):
    os.environ["LDPC_HIP_BP_PATH"] = path
    L.reload_switches()
    dec.mc_run(p, 2024, 0, shots)
    reps = 20 if shots <= 4096 else 3
    t0 = time.perf_counter()
    for r in range(reps):
        cnt = dec.mc_run(p, 2024, 0, shots)
    return (time.perf_counter() - t0) / reps * 1e3, list(cnt)

for name, shape, p in (("C1", (1200, 900, 3, 4), 0.02), ("e2100", (700, 525, 3, 4), 0.02), ("e6000", (2000, 1000, 3, 6), 0.02), ("e9000", (3000, 1500, 3, 6), 0.02), ("e12000", (4000, 2000, 3, 6), 0.02), ("e18000", (6000, 3000, 3, 6), 0.02)):
    code = L.LinearCode.random_regular_code(*shape, seed=1)
    dec = L.BpDecoder(code, p, 100)
    for shots in (1, 1024, 16384, 131072, 1048576):
        th, ch = run(dec, p, shots, "hbm")
        tl, cl = run(dec, p, shots, "lds")
        assert ch == cl, (ch, cl)
        print("%-6s E=%5d shots=%8d hbm forms %9.3f ms  lds form %9.3f ms  (x%.2f)" % (name, code.num_edges(), shots, th, tl, th / tl), flush=True)
