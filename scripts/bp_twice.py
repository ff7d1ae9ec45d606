"""diagnostics: does the BP rate depend on the size of the call, or on how long the GPU has been busy?
usage: bp_twice.py [log2 sizes ...]"""
import sys
sys.path.insert(0, ".")
import ldpc_amd as L
code = L.LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
dec = L.BpDecoder(code, 0.02, 100)
dec.mc_run(0.02, 2024, 1 << 40, 8192)
sizes = [int(x) for x in sys.argv[1:]] or [19, 21, 23, 19, 21, 19]
prev = None
for lg in sizes:
    n = 1 << lg
    cnt = dec.mc_run(0.02, 2024, 0, n)
    ms = L.last_kernel_ms()
    line = "2^%d ms %.1f GB/s %.0f capped %d" % (lg, ms, int(cnt[4]) * 960e3 / ms / 1e6, int(cnt[5]))
    if prev and n > prev[0]:
        line += "  incremental GB/s %.0f" % ((int(cnt[4]) - prev[2]) * 960e3 / (ms - prev[1]) / 1e6)
    prev = (n, ms, int(cnt[4]))
    print(line, flush=True)
