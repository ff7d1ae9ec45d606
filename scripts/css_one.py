"""One CSS BP Monte-Carlo launch pair (diagnostics).  usage: css_one.py [shots] [p]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd
if os.environ.get("AB_LIB"):   # A/B aid: another build of the library
    ldpc_amd.LIB_PATH = os.environ["AB_LIB"]
from ldpc_amd import CssCode, CssDecoder, LinearCode
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
base = LinearCode.random_regular_code(20, 15, 3, 4, seed=1)
css = CssCode.hypergraph_product(base, base)
dec = CssDecoder(css, p, 100)
dec.mc_run(p, 1, 0, 4096)
cnt = dec.mc_run(p, 1, 0, shots)
ms = ldpc_amd.last_kernel_ms()
print("css shots", shots, "ms", ms, "shots/s %.3g" % (shots / ms * 1e3), "iters/shot", int(cnt[4]) / shots, "fail", int(cnt[1]),
      "ran/shot", ldpc_amd.last_bp_executed_iterations() / shots)
