"""Throughput probe of the BP kernel (diagnostics, not a test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ldpc_amd
from ldpc_amd import BpDecoder, CssCode, CssDecoder, LinearCode

def run(name, dec, p, shots, E):
    dec.mc_run(p, 1, 0, min(shots, 4096))
    cnt = dec.mc_run(p, 1, 0, shots)
    ms = ldpc_amd.last_kernel_ms()
    iters = int(cnt[4])
    gbs = iters * 32 * E / (ms * 1e-3) / 1e9
    print("%s shots=%d  %.2f ms  %.3g shots/s  iters/shot=%.2f  fail=%d  algorithmic %.1f GB/s (%.1f%% of 8 TB/s)" %
          (name, shots, ms, shots / ms * 1e3, iters / shots, int(cnt[1]), gbs, gbs / 80), flush=True)

c1 = LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)
run("C1 bp p=0.02", BpDecoder(c1, 0.02, 100), 0.02, 1 << 21, 3600)
c2 = LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
run("C2 bp p=0.02", BpDecoder(c2, 0.02, 100), 0.02, 1 << 19, 30000)
run("C2 bp p=0.05", BpDecoder(c2, 0.05, 100), 0.05, 1 << 20, 30000)
base = LinearCode.random_regular_code(20, 15, 3, 4, seed=1)
css = CssCode.hypergraph_product(base, base)
dec = CssDecoder(css, 0.01, 100)
dec.mc_run(0.01, 1, 0, 4096)
cnt = dec.mc_run(0.01, 1, 0, 1 << 20)
ms = ldpc_amd.last_kernel_ms()
print("C3 css bp p=0.01 shots=%d %.2f ms %.3g shots/s iters(x+z)/shot=%.2f fail=%d algorithmic %.1f GB/s" %
      (1 << 20, ms, (1 << 20) / ms * 1e3, int(cnt[4]) / (1 << 20), int(cnt[1]), int(cnt[4]) * 32 * 2100 / (ms * 1e-3) / 1e9), flush=True)
if len(sys.argv) > 1:
    c5 = LinearCode.random_regular_code(1000000, 500000, 3, 6, seed=1)
    run("C5 bp p=0.02", BpDecoder(c5, 0.02, 100), 0.02, 2048, 3000000)
