#!/usr/bin/env python3
"""examples/css_decoding.rs:7-45 on the MI355X: hypergraph product of a random
(3,4)-regular n=20, m=15 code (625 qubits), depolarizing p=0.01, CssDecoder<BpDecoder>
with 100 iterations, Monte-Carlo loop on device.  Prints the number of failures."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldpc_amd import CNT_FAILURES, CssCode, CssDecoder, DepolarizingNoise, LinearCode, Probability


def code(seed):
    base = LinearCode.random_regular_code(num_bits=20, num_checks=15, bit_degree=3, check_degree=4, seed=seed)
    return CssCode.hypergraph_product(base, base)


def main():
    max_iterations = 100
    probability = Probability(0.01)
    shots = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    css = code(seed=1)
    decoder = CssDecoder(css, probability, max_iterations)
    noise = DepolarizingNoise.with_probability(probability)
    # one shot through the reference's own call sequence (css_decoding.rs:15-18) ...
    positions, paulis = css.random_error(noise, seed=2024, shot=0)
    x_part = [q for q, p in zip(positions, paulis) if p in "XY"]
    z_part = [q for q, p in zip(positions, paulis) if p in "ZY"]
    sx, sz = css.syndrome_of(x_part, z_part)
    cx, cz = decoder.correction_for(sx, sz)
    ok = css.has_stabilizer(sorted(set(x_part) ^ set(cx)), sorted(set(z_part) ^ set(cz)))
    # ... and the whole loop fused on device
    counters = decoder.mc_run(probability, seed=2024, first_shot=0, n_shots=shots)
    print("single shot corrected:", ok, file=sys.stderr)
    print(int(counters[CNT_FAILURES]))


if __name__ == "__main__":
    main()
