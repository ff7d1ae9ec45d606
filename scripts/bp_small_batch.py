"""Persistent vs streamed BP on small batches (few 64-shot groups): where is the crossover?"""
import os
import sys
import time

sys.path.insert(0, ".")
import ldpc_amd as L


def run(dec, p, shots, path):
    if path:
        os.environ["LDPC_HIP_BP_PATH"] = path
        L.reload_switches()
    else:
        os.environ.pop("LDPC_HIP_BP_PATH", None)
        L.reload_switches()
    dec.mc_run(p, 2024, 0, shots)
    reps = 5 if shots <= 64 * 64 else 2
    t0 = time.perf_counter()
    for r in range(reps):
        cnt = dec.mc_run(p, 2024, 0, shots)
    return (time.perf_counter() - t0) / reps * 1e3, list(cnt)


def main():
    cases = [("C1", (1200, 900, 3, 4), 0.02), ("mid", (4000, 2000, 3, 6), 0.02), ("C2", (10000, 5000, 3, 6), 0.02),
             ("big", (40000, 20000, 3, 6), 0.02)]
    if len(sys.argv) > 1 and sys.argv[1] == "small":
        cases = [("ham", None, 0.1), ("e60", (20, 15, 3, 4), 0.05), ("e240", (80, 60, 3, 4), 0.03), ("e600", (200, 150, 3, 4), 0.02),
                 ("e1200", (400, 300, 3, 4), 0.02), ("e2100", (700, 525, 3, 4), 0.02)]
    for name, shape, p in cases:
        code = L.LinearCode.random_regular_code(*shape, seed=1) if shape else L.LinearCode.hamming_code()
        dec = L.BpDecoder(code, p, 100)
        for groups in (1, 4, 16, 32, 64, 128, 256, 512):
            shots = groups * 64
            tp, cp = run(dec, p, shots, "persistent")
            ts, cs = run(dec, p, shots, "stream")
            td, cd = run(dec, p, shots, None)
            assert cp == cs == cd, (cp, cs, cd)
            print("%-4s E=%6d groups=%4d persistent %9.3f ms  stream %9.3f ms  default %9.3f ms" % (name, code.num_edges(), groups, tp, ts, td), flush=True)


if __name__ == "__main__":
    main()
