"""diagnostics: does the BP rate depend on the size of the call, or on how long the GPU has been busy?"""
import sys
sys.path.insert(0, ".")
import ldpc_amd as L
code = L.LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
dec = L.BpDecoder(code, 0.02, 100)
dec.mc_run(0.02, 2024, 1 << 40, 8192)
for lg in (19, 21, 23, 19, 21, 19):
    n = 1 << lg
    cnt = dec.mc_run(0.02, 2024, 0, n)
    ms = L.last_kernel_ms()
    print("2^%d" % lg, "ms %.1f" % ms, "GB/s %.0f" % (int(cnt[4]) * 960e3 / ms / 1e6), "capped", int(cnt[5]), flush=True)
