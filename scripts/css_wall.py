"""CSS Monte-Carlo: wall clock around the call next to the library's kernel time.  usage: css_wall.py [shots]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd
from ldpc_amd import CssCode, CssDecoder, LinearCode
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 23
base = LinearCode.random_regular_code(20, 15, 3, 4, seed=1)
css = CssCode.hypergraph_product(base, base)
dec = CssDecoder(css, 0.01, 100)
dec.mc_run(0.01, 1, 0, 4096)
for rep in range(2):
    t0 = time.perf_counter()
    cnt = dec.mc_run(0.01, 1, 0, shots)
    wall = (time.perf_counter() - t0) * 1e3
    print("shots", shots, "wall ms %.1f" % wall, "kernel ms %.1f" % ldpc_amd.last_kernel_ms(), "shots/s %.3g" % (shots / wall * 1e3), list(map(int, cnt)), flush=True)
