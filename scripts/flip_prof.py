"""Flip Monte-Carlo at C2 with the kernel's phase stamps (LDPC_HIP_FLIP_PROF=1; diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd
if os.environ.get("AB_LIB"):   # a variant library (scripts/build_variant.sh)
    ldpc_amd.LIB_PATH = os.environ["AB_LIB"]
from ldpc_amd import FlipDecoder, LinearCode
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
code = LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
dec = FlipDecoder(code)
dec.mc_run(p, 1, 0, 4096)
best = 1e9
for rep in range(3):
    c = dec.mc_run(p, 2024, 0, shots)
    best = min(best, ldpc_amd.last_kernel_ms())
print("kernel_ms %.3f shots %d -> %.3e shots/s; flips/shot %.1f" % (best, shots, shots / best * 1e3, int(c[4]) / shots))
if os.environ.get("LDPC_HIP_FLIP_PROF") and os.environ.get("LDPC_HIP_FLIP_KERNEL") != "v3" and shots >= 65536:
    it, live = int(c[5]) & 0xFFFFFFFF, int(c[5]) >> 32
    ticks = int(c[6])
    busy, tot = (int(c[7]) & 0xFFFFFFFF) << 10, (int(c[7]) >> 32) << 10
    print("flip waves: %d passes, %.2f live groups per pass, %.1f ticks per pass; service waves busy %.3f of their time, %.0f ticks per shot" %
          (it, live / max(it, 1), ticks / max(it, 1), busy / max(tot, 1), busy / shots))
elif os.environ.get("LDPC_HIP_FLIP_PROF"):
    it, tl, ts = int(c[5]), int(c[6]), int(c[7])
    print("wave iterations %d (%.1f per shot-group of 8 -> x8/shots = %.1f), loop ticks/iteration %.1f, loop share %.3f" %
          (it, it * 8 / shots, it * 8 / shots, tl / max(it, 1), tl / max(tl + ts, 1)))
