#!/usr/bin/env python3
"""examples/random_linear_code.rs:5-17: sample a (3,4)-regular n=4, m=3 code and one
BSC(0.2) error (drawn on the GPU), print the error."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldpc_amd import BinarySymmetricChannel, LinearCode, Probability

code = LinearCode.random_regular_code(num_bits=4, num_checks=3, bit_degree=3, check_degree=4, seed=1)
noise = BinarySymmetricChannel.with_probability(Probability(0.2))
error = code.random_error(noise, seed=2024, shot=0)
print(error)
