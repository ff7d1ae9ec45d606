"""The oracle against the kernels that bench.py TIMES, reached the way the bench reaches them.

A call that asks for posteriors or check messages is served by bp_kernel / bps_* (they keep both message arrays), so
the oracle tests of test_gpu_bp.py / test_gpu_bp_wide.py that pass `dp` never run bp_lds_kernel, which is what the
`bp`, `css` and `bp_c1` lines of the bench time.  Here every call is a plain one (corrections + iteration counts only),
the test asserts which kernel served it (ldpc_last_kernel_name), and the corrections and iteration counts are compared
with the oracle's restatement of belief_propagation.rs:23-30, 150-180 on every shot: decisions >= 99.99 % on the shots
that converge, iteration counts equal wherever decisions are, every differing shot followed through its first three
iterations message by message (tests/gpu_util.py::assert_bp_matches_oracle).

Last, configs[4]: four shots of the streamed form (bps_iterate<6,3>) at n = 10^6 against the oracle -- corrections,
iteration counts and posteriors at rtol 1e-5 -- at the size where its 32-bit offsets and 3 GB groups are most at risk."""
import numpy as np
import pytest

import ldpc_amd
from ldpc_amd import BinarySymmetricChannel, BpDecoder, CssCode, LinearCode, nwords
from tests.gpu_util import (agree_f64, assert_bp_matches_oracle, dev_f64, dev_u32, dev_u64, host_u32, host_u64, oracle_code)

pytestmark = pytest.mark.gpu

SEED = 2024


def _plain_call(code, p, cap, shots, seed=SEED):
    """sample + syndrome on the device, then correction_for_batch WITHOUT posterior / message outputs"""
    n, m = code.len(), code.num_checks()
    de, ds = dev_u64((shots, nwords(n))), dev_u64((shots, nwords(m)))
    BinarySymmetricChannel(p).sample_batch(n, seed, 0, shots, de)
    code.syndrome_batch(shots, de, ds)
    dc, di = dev_u64((shots, nwords(n))), dev_u32((shots,))
    BpDecoder(code, p, cap).correction_for_batch(shots, ds, dc, di)
    return host_u64(ds), host_u64(dc), host_u32(di), ldpc_amd.last_kernel_name()


def test_c2_hybrid_form_against_the_oracle():
    # configs[1]: the bench's `bp` line times bp_lds_kernel<6,3,wide,hybrid>
    code = LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
    syn, gc, gi, kernel = _plain_call(code, 0.02, 100, 10240)
    assert kernel.startswith("bp_lds_kernel") and "hybrid" in kernel, kernel
    share = assert_bp_matches_oracle(code, oracle_code(code), 0.02, 100, syn, gc, gi)
    assert share >= 0.9999, share


def test_c1_lds_form_against_the_oracle():
    # configs[0]'s code: the bench's `bp_c1` line times bp_lds_kernel<4,3>
    code = LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)
    syn, gc, gi, kernel = _plain_call(code, 0.02, 100, 10240)
    assert kernel.startswith("bp_lds_kernel") and "hybrid" not in kernel, kernel
    share = assert_bp_matches_oracle(code, oracle_code(code), 0.02, 100, syn, gc, gi)
    assert share >= 0.9999, share


def test_css_half_degree7_form_against_the_oracle():
    # configs[2]: one half of the CSS example (Hx of the hypergraph product: check degree 7, bit degrees 3 and 4), the
    # bench's `css` line times bp_lds_kernel<7,-4>.  About 5 % of the shots never converge at p = 0.01 on this
    # degenerate code (period-two runs, ended early by the kernel, NaN / inf messages): they count towards the
    # overall bar only, the converging ones towards 99.99 %.
    base = LinearCode.random_regular_code(20, 15, 3, 4, seed=1)
    hx = CssCode.hypergraph_product(base, base).x_stabs_binary()
    syn, gc, gi, kernel = _plain_call(hx, 0.01, 100, 10240)
    assert kernel.startswith("bp_lds_kernel<7"), kernel
    assert (gi == 100).sum() > 50          # the runs at the cap are there
    assert_bp_matches_oracle(hx, oracle_code(hx), 0.01, 100, syn, gc, gi)


def test_c5_streamed_form_against_the_oracle():
    # configs[4]: bps_iterate<6,3> at n = 10^6 (E = 3e6: one oracle shot is ~1e8 libm calls, seconds)
    n, m, shots, p = 1000000, 500000, 4, 0.02
    code = LinearCode.random_regular_code(n, m, 3, 6, seed=1)
    oc = oracle_code(code)
    de, ds = dev_u64((shots, nwords(n))), dev_u64((shots, nwords(m)))
    BinarySymmetricChannel(p).sample_batch(n, SEED, 0, shots, de)
    code.syndrome_batch(shots, de, ds)
    syn = host_u64(ds)
    dec = BpDecoder(code, p, 100)
    # plain call (what the bench's bp_c5 line runs) and the call with posteriors
    dc, di = dev_u64((shots, nwords(n))), dev_u32((shots,))
    dec.correction_for_batch(shots, ds, dc, di)
    assert ldpc_amd.last_kernel_name().startswith("bps_"), ldpc_amd.last_kernel_name()
    dc2, di2, dp = dev_u64((shots, nwords(n))), dev_u32((shots,)), dev_f64((shots, n))
    dec.correction_for_batch(shots, ds, dc2, di2, dp)
    assert ldpc_amd.last_kernel_name().startswith("bps_"), ldpc_amd.last_kernel_name()
    rc, ri, rp, _ = oc.bp_correction_for_batch(p, 100, syn, want_post=True, nthreads=4)
    gc, gi = host_u64(dc), host_u32(di)
    assert (gc == rc).all() and (gi == ri).all(), (gi, ri)
    assert (host_u64(dc2) == rc).all() and (host_u32(di2) == ri).all()
    assert agree_f64(dp.cpu().numpy(), rp).all()
    assert (rc == host_u64(de)).all() and (ri >= 1).all() and (ri < 100).all()   # (and the correction is the error)
