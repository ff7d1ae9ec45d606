"""Latency of the per-shot drop-in calls (the granularity of the crate's traits): ldpc_bp_correction_for,
ldpc_flip_decode, ldpc_css_bp_correction_for, ldpc_code_syndrome_of.  One call = host position lists in, one
kernel launch, host position list out."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import ldpc_amd as L


def timed(label, fn, reps=500):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = (time.perf_counter() - t0) / reps
    print("%-52s %8.1f us/call" % (label, dt * 1e6), flush=True)


def main():
    rng = np.random.default_rng(5)
    for name, (n, m, dv, dc) in (("C1 (3,4) n=1200", (1200, 900, 3, 4)), ("C2 (3,6) n=10000", (10000, 5000, 3, 6))):
        code = L.LinearCode.random_regular_code(n, m, dv, dc, seed=1)
        err = sorted(rng.choice(n, size=max(1, int(0.02 * n)), replace=False).tolist())
        synd = code.syndrome_of(err)
        bp = L.BpDecoder(code, 0.02, 100)
        flip = L.FlipDecoder(code)
        timed(name + " syndrome_of", lambda: code.syndrome_of(err))
        timed(name + " BpDecoder.correction_for", lambda: bp.correction_for(synd))
        timed(name + " FlipDecoder.decode", lambda: flip.decode(err))
    base = L.LinearCode.random_regular_code(20, 15, 3, 4, seed=1)
    css = L.CssCode.hypergraph_product(base, base)
    dec = L.CssDecoder(css, 0.01, 100)
    noise = L.DepolarizingNoise(0.01)
    pos, paulis = noise.sample_error_of_length(css.len(), 2024, 3)
    xs = [p for p, k in zip(pos, paulis) if k in ('X', 'Y')]
    zs = [p for p, k in zip(pos, paulis) if k in ('Z', 'Y')]
    sx, sz = css.syndrome_of(x_part=xs, z_part=zs)
    timed("C3 HGP 625 qubits CssDecoder.correction_for", lambda: dec.correction_for(sx, sz))
    timed("C3 depolarizing sample_error_of_length", lambda: noise.sample_error_of_length(css.len(), 2024, 3))


if __name__ == "__main__":
    main()
