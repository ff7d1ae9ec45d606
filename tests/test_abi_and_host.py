"""CPU-only checks of the boundary: the C-ABI library loads, exports every symbol
declared in include/ldpc_hip.h, and the host-side code construction (which needs
no GPU) keeps the reference's contracts.  No kernel is launched here."""
import os
import re

import numpy as np
import pytest

import ldpc_amd
from ldpc_amd import CssCode, FlipDecoder, LinearCode
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "ldpc_hip.h")).read()
    declared = set(re.findall(r"^(?:const char \*|const ldpc_code \*|int |size_t |uint64_t |double |void )(ldpc_[a-z0-9_]+)\(",
                              header, flags=re.M))
    assert len(declared) >= 45
    L = ldpc_amd.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "missing export " + name
    assert declared == set(L._signatures), "python binding table out of sync with the header"


def test_hamming_and_edges(kats):
    code = LinearCode.hamming_code()
    assert code.len() == 7 and code.num_checks() == 3 and code.num_edges() == 12
    assert code.parity_check_matrix() == kats["hamming"]["rows"]
    assert code.edges() == [tuple(e) for e in kats["hamming_edges"]["edges_bit_check"]]
    d = kats["edges_doctest"]
    c2 = LinearCode.from_parity_check_matrix(d["n"], d["rows"])
    assert c2.edges() == [tuple(e) for e in d["edges_bit_check"]]
    # bit_adjacencies = transposed (linear_code/mod.rs:92), rows ascending
    assert code.bit_adjacencies() == orc.Code(7, rows=kats["hamming"]["rows"]).bit_adjacencies()


def test_repetition_code():
    code = LinearCode.repetition_code(3)
    assert code.parity_check_matrix() == [[0, 1], [1, 2]]


def test_invalid_matrices_rejected():
    for rows in ([[1, 0]], [[0, 4]], [[1, 1]]):
        with pytest.raises(ldpc_amd.LdpcError) as e:
            LinearCode.from_parity_check_matrix(4, rows)
        assert e.value.status == ldpc_amd.ERR_BAD_ARG


def test_random_regular_contract(kats):
    d = kats["random_regular_contract"]
    code = LinearCode.random_regular_code(d["num_bits"], d["num_checks"], d["bit_degree"], d["check_degree"], seed=3)
    assert code.len() == d["num_bits"] and code.num_checks() == d["num_checks"]
    assert code.num_edges() == d["number_of_ones"]
    for row in code.parity_check_matrix():
        assert len(row) == d["check_degree"] and row == sorted(set(row))
    for col in code.bit_adjacencies():
        assert len(col) == d["bit_degree"] and col == sorted(set(col))
    # n*dv != m*dc -> SamplingError (random.rs:52-65)
    with pytest.raises(ldpc_amd.SamplingError):
        LinearCode.random_regular_code(20, 15, 3, 5, seed=3)
    # deterministic in the seed, different across seeds
    again = LinearCode.random_regular_code(d["num_bits"], d["num_checks"], d["bit_degree"], d["check_degree"], seed=3)
    other = LinearCode.random_regular_code(d["num_bits"], d["num_checks"], d["bit_degree"], d["check_degree"], seed=4)
    assert again.parity_check_matrix() == code.parity_check_matrix()
    assert other.parity_check_matrix() != code.parity_check_matrix()


@pytest.mark.parametrize("n,m,dv,dc", [(4, 3, 3, 4), (1200, 900, 3, 4), (10000, 5000, 3, 6), (6, 6, 5, 5), (0, 0, 0, 0)])
def test_random_regular_shapes(n, m, dv, dc):
    # (4,3,3,4) is the complete graph of examples/random_linear_code.rs:6-12
    code = LinearCode.random_regular_code(n, m, dv, dc, seed=1)
    ptr, idx = code.csr()
    assert len(ptr) == m + 1 and len(idx) == n * dv
    if m:
        assert (np.diff(ptr.astype(np.int64)) == dc).all()
        rows = idx.reshape(m, dc).astype(np.int64)
        assert (np.diff(rows, axis=1) > 0).all()             # simple graph, neighbours ascending
        assert (np.bincount(idx.astype(np.int64), minlength=n) == dv).all()


def test_large_code_builds_without_generator_matrix():
    code = LinearCode.random_regular_code(1_000_000, 500_000, 3, 6, seed=1)
    assert code.num_edges() == 3_000_000


def test_shor_and_steane(kats):
    shor = CssCode.shor_code()
    d = kats["shor"]
    assert shor.len() == 9
    assert shor.x_stabs_binary().parity_check_matrix() == d["hx"]
    assert shor.z_stabs_binary().parity_check_matrix() == d["hz"]
    assert shor.x_logicals_binary().parity_check_matrix() == d["lx"]
    assert shor.z_logicals_binary().parity_check_matrix() == d["lz"]
    st = kats["steane_stabilizers"]
    steane = CssCode.steane_code()
    assert steane.x_stabs_binary().parity_check_matrix() == st["x_rows"]
    assert steane.z_stabs_binary().parity_check_matrix() == st["z_rows"]


def _check_logicals(css):
    n = css.len()
    hx = orc.dense(css.x_stabs_binary().parity_check_matrix(), n).astype(int)
    hz = orc.dense(css.z_stabs_binary().parity_check_matrix(), n).astype(int)
    lx = orc.dense(css.x_logicals_binary().parity_check_matrix(), n).astype(int)
    lz = orc.dense(css.z_logicals_binary().parity_check_matrix(), n).astype(int)
    assert not ((hx @ hz.T) % 2).any()
    # what logicals.rs:139-158 means to assert: logicals commute with the
    # stabilizers of the other kind and pair up anticommuting
    assert not ((lx @ hz.T) % 2).any()
    assert not ((lz @ hx.T) % 2).any()
    assert (((lx @ lz.T) % 2) == np.eye(lx.shape[0], dtype=int)).all()
    k = n - orc.gf2_rank(hx.astype(np.uint8)) - orc.gf2_rank(hz.astype(np.uint8))
    assert lx.shape[0] == k and lz.shape[0] == k
    return k


def test_logicals_steane_toric_hgp(kats):
    assert _check_logicals(CssCode.steane_code()) == 1
    assert _check_logicals(CssCode.toric_code(3)) == 2
    d = kats["hgp_repetition3"]
    rep = LinearCode.repetition_code(3)
    surface = CssCode.hypergraph_product(rep, rep)
    assert surface.len() == d["num_qubits"]
    assert _check_logicals(surface) == 1
    # block layout against the numpy restatement of css/mod.rs:121-146
    hx_rows, hz_rows, nq = orc.hypergraph_product(d["repetition_rows"], 3, d["repetition_rows"], 3)
    assert surface.x_stabs_binary().parity_check_matrix() == hx_rows
    assert surface.z_stabs_binary().parity_check_matrix() == hz_rows


def test_hgp_of_random_code_matches_example_shape():
    # examples/css_decoding.rs:25-34
    base = LinearCode.random_regular_code(20, 15, 3, 4, seed=1)
    css = CssCode.hypergraph_product(base, base)
    assert css.len() == 625 and css.num_x_stabs() == 300 and css.num_z_stabs() == 300
    rows = base.parity_check_matrix()
    hx_rows, hz_rows, _ = orc.hypergraph_product(rows, 20, rows, 20)
    assert css.x_stabs_binary().parity_check_matrix() == hx_rows
    assert css.z_stabs_binary().parity_check_matrix() == hz_rows
    assert all(len(r) == 7 for r in hx_rows)
    _check_logicals(css)


def test_css_errors():
    a = LinearCode.repetition_code(3)
    b = LinearCode.repetition_code(4)
    with pytest.raises(ldpc_amd.CssError):
        CssCode.new(a, b)                       # CssError::DifferentXandZLength
    with pytest.raises(ldpc_amd.CssError):
        CssCode.new(a, a)                       # CssError::NonOrthogonalCodes


def test_probability_range():
    with pytest.raises(ValueError):
        ldpc_amd.Probability(1.5)
    assert ldpc_amd.Probability.try_new(-0.1) is None
    assert ldpc_amd.Probability(0.25).value() == 0.25


def test_compute_fails_loudly_without_gpu():
    if ldpc_amd.device_count() > 0:
        pytest.skip("GPU present")
    code = LinearCode.hamming_code()
    with pytest.raises(ldpc_amd.LdpcError) as e:
        code.syndrome_of([0, 2, 4])
    assert e.value.status == ldpc_amd.ERR_HIP


def test_probabilities_the_samplers_cannot_represent_are_refused():
    """noise/mod.rs:35-46 accepts any p in [0, 1] and rand's Bernoulli compares 64 bits; the device samplers hold
    P(gap >= j) in 32 bits, so 0 < p < 1e-7 would be sampled at a visibly different rate: refused with
    LDPC_ERR_BAD_ARG before anything is launched.  A decoder PRIOR below that stays legal (it is not sampled)."""
    import ctypes as C
    L = ldpc_amd.lib()
    code = LinearCode.hamming_code()
    dec = FlipDecoder(code)
    cnt = (C.c_uint64 * 8)()
    for p in (1e-9, 9.9e-8, 2.0 ** -40):
        assert L.ldpc_flip_mc_run(dec._h, p, 1, 0, 4, cnt, None, None, None) == ldpc_amd.ERR_BAD_ARG
        assert b"1e-7" in L.ldpc_last_error()
        assert L.ldpc_bsc_sample_batch(p, 64, 1, 0, 1, None) == ldpc_amd.ERR_BAD_ARG
    from ldpc_amd import BpDecoder
    BpDecoder(code, 1e-12, 5)      # prior only: accepted


def test_code_create_rejects_a_missing_column_array():
    import ctypes as C
    L = ldpc_amd.lib()
    ptr = (C.c_size_t * 2)(0, 2)
    out = C.c_void_p()
    assert L.ldpc_code_create(4, 1, ptr, None, C.byref(out)) == ldpc_amd.ERR_BAD_ARG
