"""BASELINE.json's configurations at their full shot counts on one GPU (Monte-Carlo entry points, device noise):
C1 1e4 shots, C2 1e8 shots (flip and BP), C3 1e7 shots (CSS BP).  Prints one JSON line per run."""
import json
import sys
import time

sys.path.insert(0, ".")
import ldpc_amd as L


def run(label, fn, shots):
    t0 = time.perf_counter()
    cnt = fn()
    dt = time.perf_counter() - t0
    print(json.dumps({"run": label, "shots": shots, "wall_s": round(dt, 3), "kernel_ms": round(L.last_kernel_ms(), 2),
                      "shots_per_s": shots / dt, "counters": {k: int(v) for k, v in zip(L.COUNTER_NAMES, cnt)}}), flush=True)


def main():
    seed = 2024
    c1 = L.LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)
    c2 = L.LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
    base = L.LinearCode.random_regular_code(20, 15, 3, 4, seed=1)
    css = L.CssCode.hypergraph_product(base, base)
    f1, b1 = L.FlipDecoder(c1), L.BpDecoder(c1, 0.02, 100)
    f2, b2 = L.FlipDecoder(c2), L.BpDecoder(c2, 0.02, 100)
    d3 = L.CssDecoder(css, 0.01, 100)
    for d in (f1, b1, f2, b2):
        d.mc_run(0.02, seed, 1 << 40, 4096)
    d3.mc_run(0.01, seed, 1 << 40, 4096)
    run("C1 flip p=0.02", lambda: f1.mc_run(0.02, seed, 0, 10000), 10000)
    run("C1 bp p=0.02", lambda: b1.mc_run(0.02, seed, 0, 10000), 10000)
    run("C2 flip p=0.02", lambda: f2.mc_run(0.02, seed, 0, 100000000), 100000000)
    run("C3 css bp depolarizing p=0.01", lambda: d3.mc_run(0.01, seed, 0, 10000000), 10000000)
    run("C2 bp p=0.02", lambda: b2.mc_run(0.02, seed, 0, 100000000), 100000000)


if __name__ == "__main__":
    main()
