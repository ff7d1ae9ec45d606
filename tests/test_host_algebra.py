"""Host algebra either side of the decoding path: LinearCode::{dimension, num_generators,
minimal_distance, has_same_codespace} and CssErasureDecoder, against the reference's own doc-tests / unit tests
(cited per case) and against numpy GF(2) elimination (oracle.gf2_rank).  No GPU needed."""
import itertools

import numpy as np
import pytest

import ldpc_amd as L
from ldpc_amd import CssCode, CssErasureDecoder, LinearCode
from oracle import oracle as orc

HAMMING = [[0, 1, 2, 4], [0, 1, 3, 5], [0, 2, 3, 6]]           # linear_code/mod.rs:255-259, 300-305


def _dense(rows, n):
    a = np.zeros((len(rows), n), dtype=np.uint8)
    for r, row in enumerate(rows):
        a[r, row] = 1
    return a


def test_dimension_and_same_codespace_doc_tests():
    ham = LinearCode.from_parity_check_matrix(7, HAMMING)
    assert ham.dimension() == 4 and ham.num_generators() == 4       # mod.rs:300-307
    other = LinearCode.from_parity_check_matrix(7, [[0, 1, 2, 4], [2, 3, 4, 5], [1, 3, 4, 6]])
    assert ham.has_same_codespace(other)                            # mod.rs:252-270
    # as written (mod.rs:272-275) the test is H_self * G_other^T == 0, i.e. other's codewords lie in self's
    # codespace: the all-ones word of the repetition code is a Hamming codeword, the converse fails
    assert ham.has_same_codespace(LinearCode.repetition_code(7))
    assert not LinearCode.repetition_code(7).has_same_codespace(ham)
    assert not ham.has_same_codespace(LinearCode.repetition_code(5))
    assert ham.minimal_distance() == 3
    assert LinearCode.repetition_code(5).minimal_distance() == 5 and LinearCode.repetition_code(5).dimension() == 1
    # generator rows are codewords and span the right dimension
    g = ham.generator_matrix()
    assert len(g) == 4 and all(ham_row_ok(g_row) for g_row in g)
    assert orc.gf2_rank(_dense(g, 7)) == 4


def ham_row_ok(row):
    h = _dense(HAMMING, 7)
    v = np.zeros(7, dtype=np.uint8)
    v[row] = 1
    return not (h @ v % 2).any()


def test_no_codeword_and_full_space():
    full_rank = LinearCode.from_parity_check_matrix(3, [[0], [1], [2]])
    assert full_rank.dimension() == 0 and full_rank.minimal_distance() is None and full_rank.generator_matrix() == []
    free = LinearCode.from_parity_check_matrix(4, [])
    assert free.dimension() == 4 and free.minimal_distance() == 1


@pytest.mark.parametrize("seed", range(6))
def test_random_codes_against_numpy(seed):
    rng = np.random.default_rng(seed)
    n, m = int(rng.integers(6, 15)), int(rng.integers(2, 9))
    rows = [sorted(rng.choice(n, size=int(rng.integers(1, min(n, 5))), replace=False).tolist()) for _ in range(m)]
    code = LinearCode.from_parity_check_matrix(n, rows)
    h = _dense(rows, n)
    k = n - orc.gf2_rank(h)
    assert code.dimension() == k == code.num_generators()
    best = None
    for bits in itertools.product((0, 1), repeat=n):                # every word: brute-force minimum distance
        v = np.array(bits, dtype=np.uint8)
        if v.any() and not (h @ v % 2).any():
            w = int(v.sum())
            best = w if best is None else min(best, w)
    assert code.minimal_distance() == best


# decoders/css_erasure.rs:73-195: (code, erasure, bad x, bad z, recovery probability)
ERASURE_KATS = [
    ("shor", [0, 4, 8], 0, 1, 0.5), ("shor", [0, 1, 2], 1, 0, 0.5),                       # :73-89
    ("shor", [0, 1, 3, 4], 0, 0, 1.0), ("shor", [0, 6, 7], 0, 0, 1.0),                    # :92-108
    ("shor", [], 0, 0, 1.0),                                                              # :111-120
    ("steane", [0, 1, 2], 1, 1, 0.25), ("steane", [0, 3, 4, 5], 1, 1, 0.25),              # :123-139
    ("toric3", [0, 1, 2], 0, 1, 0.5), ("toric3", [9, 10, 11], 1, 0, 0.5),                 # :142-155
    ("toric3", [1, 4, 6, 7, 8, 9, 12, 15], 1, 2, 0.125),                                  # :157-161
    ("toric3", [1, 4, 6, 7, 8, 9, 10, 11, 12, 15], 2, 2, 0.0625),                         # :163-167
    ("toric3", [0, 1], 0, 0, 1.0), ("toric3", [0, 1, 3, 4, 9, 10, 12, 13], 0, 0, 1.0),    # :170-194
]


@pytest.mark.parametrize("name,erasure,bad_x,bad_z,prob", ERASURE_KATS)
def test_css_erasure_decoder_kats(name, erasure, bad_x, bad_z, prob):
    code = {"shor": CssCode.shor_code, "steane": CssCode.steane_code, "toric3": lambda: CssCode.toric_code(3)}[name]()
    dec = CssErasureDecoder(code)
    assert dec.num_bad_x_errors(erasure) == bad_x
    assert dec.num_bad_z_errors(erasure) == bad_z
    assert dec.recovery_probability(erasure) == prob


def test_css_erasure_against_numpy_ranks():
    base = LinearCode.random_regular_code(8, 6, 3, 4, seed=3)
    css = CssCode.hypergraph_product(base, base)
    n = css.len()
    dec = CssErasureDecoder(css)
    mats = {k: _dense(getattr(css, k)().parity_check_matrix(), n)
            for k in ("x_stabs_binary", "z_stabs_binary", "x_logicals_binary", "z_logicals_binary")}
    rng = np.random.default_rng(1)
    for _ in range(20):
        er = sorted(rng.choice(n, size=int(rng.integers(0, n // 2)), replace=False).tolist())

        def bad(stabs, logs):
            s, l = mats[stabs][:, er].T, mats[logs][:, er].T
            if not er:
                return 0
            return orc.gf2_rank(np.concatenate([s, l], axis=1)) - orc.gf2_rank(s)

        assert dec.num_bad_x_errors(er) == bad("z_stabs_binary", "z_logicals_binary")
        assert dec.num_bad_z_errors(er) == bad("x_stabs_binary", "x_logicals_binary")
    with pytest.raises(Exception):
        dec.recovery_probability([n])          # out of range


def test_from_generator_matrix_doc_test_and_from_both_matrices():
    """linear_code/mod.rs:104-115 (3-bit repetition code from its generator), :59-72 (from_both_matrices and its
    panic on non-orthogonal matrices), :169-175 (empty)"""
    code = L.LinearCode.from_generator_matrix(3, [[0, 1, 2]])
    assert code.len() == 3 and code.dimension() == 1 and code.minimal_distance() == 3
    assert code.has_same_codespace(L.LinearCode.repetition_code(3))
    ham = L.LinearCode.hamming_code()
    gen = ham.generator_matrix()
    both = L.LinearCode.from_both_matrices(7, gen, ham.parity_check_matrix())
    assert both.parity_check_matrix() == ham.parity_check_matrix()
    back = L.LinearCode.from_generator_matrix(7, gen)                      # H = nullspace(G): same codespace
    assert back.has_same_codespace(ham) and ham.has_same_codespace(back) and back.dimension() == 4
    with pytest.raises(L.LdpcError) as e:
        L.LinearCode.from_both_matrices(7, [[0]], ham.parity_check_matrix())
    assert "non orthogonal" in str(e.value)
    with pytest.raises(L.LdpcError):
        L.LinearCode.from_generator_matrix(3, [[0, 3]])                    # column out of range
    empty = L.LinearCode.empty()
    assert empty.len() == 0 and empty.num_checks() == 0 and empty.dimension() == 0
