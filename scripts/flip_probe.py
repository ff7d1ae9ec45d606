"""Latency/throughput probe of the flip kernel (diagnostics, not a test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ldpc_amd
from ldpc_amd import FlipDecoder, LinearCode

code = LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
dec = FlipDecoder(code)
dec.mc_run(0.02, 1, 0, 1024)
for p in (0.02, 0.01, 0.001, 0.0):
    for shots in (4, 256 * 4, 2560 * 4, 2560 * 4 * 4, 2560 * 4 * 16):
        best = 1e9
        for rep in range(3):
            cnt = dec.mc_run(p, 1, 0, shots)
            best = min(best, ldpc_amd.last_kernel_ms())
        print("p=%g shots=%7d  %8.3f ms  %10.0f shots/s  flips/shot=%.1f" % (p, shots, best, shots / best * 1e3, cnt[4] / shots), flush=True)
