"""Pins the CPU oracle (oracle/ldpc_oracle.c) against every known-answer test the
reference holds for the Monte-Carlo decoding path.  CPU only."""
import numpy as np
import pytest

from oracle import oracle as orc


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10 (published algorithm, Salmon et al. SC'11)
    assert orc.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert orc.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert orc.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_edges_order(kats):
    ham = orc.Code(kats["hamming"]["n"], rows=kats["hamming"]["rows"])
    assert ham.edges() == [tuple(e) for e in kats["hamming_edges"]["edges_bit_check"]]
    d = kats["edges_doctest"]
    code = orc.Code(d["n"], rows=d["rows"])
    assert code.edges() == [tuple(e) for e in d["edges_bit_check"]]


def test_syndrome_doctest(kats):
    d = kats["syndrome_doctest"]
    code = orc.Code(d["n"], rows=d["rows"])
    assert code.syndrome_of(d["message"]) == d["syndrome"]
    assert code.syndrome_of(d["codeword"]) == []
    assert code.syndrome_of(d["not_codeword"]) != []
    with pytest.raises(ValueError):
        code.syndrome_of([0], length=8)  # reference panics on length mismatch
    # packed form agrees
    s = code.syndrome_words(orc.pos_to_words(d["message"], d["n"]))
    assert orc.words_to_pos(s, code.m) == d["syndrome"]


def test_flip_hamming(kats):
    ham = orc.Code(kats["hamming"]["n"], rows=kats["hamming"]["rows"])
    for case in kats["flip_hamming"]["cases"]:
        out, _ = ham.flip_decode(case["in"])
        assert out == case["out"]
        w, _, _ = ham.flip_decode_words(orc.pos_to_words(case["in"], 7))
        assert orc.words_to_pos(w, 7) == case["out"]


def test_bp_hamming(kats):
    ham = orc.Code(kats["hamming"]["n"], rows=kats["hamming"]["rows"])
    d = kats["bp_hamming"]
    side = kats["session_derived_bp_side_values"]
    for i, case in enumerate(d["cases"]):
        msg = orc.pos_to_words(case["in"], 7)
        out, iters = ham.bp_decode_words(d["probability"], d["max_iterations"], msg)
        assert orc.words_to_pos(out, 7) == case["out"]
        assert iters == side["iterations"][i]
        corr, it2, _ = ham.bp_correction_for(d["probability"], d["max_iterations"], ham.syndrome_words(msg))
        assert orc.words_to_pos(corr, 7) == side["corrections"][i]
        assert it2 == iters


def test_bp_first_check_message_magnitude(kats):
    ham = orc.Code(kats["hamming"]["n"], rows=kats["hamming"]["rows"])
    syn = ham.syndrome_words(orc.pos_to_words([0], 7))
    _, _, _, chk = ham.bp_correction_for(0.1, 1, syn, want_messages=True)
    ref = kats["session_derived_bp_side_values"]["first_check_message_magnitude_p0.1_dc4"]
    assert np.allclose(np.abs(chk), ref, rtol=1e-11)


def test_bp_max_iter_zero_returns_prior_decision(kats):
    # belief_propagation.rs:26-27: the stop test runs before the first update
    ham = orc.Code(kats["hamming"]["n"], rows=kats["hamming"]["rows"])
    syn = ham.syndrome_words(orc.pos_to_words([0], 7))
    corr, it, post = ham.bp_correction_for(0.1, 0, syn)
    assert it == 0 and not corr.any()
    corr, it, post = ham.bp_correction_for(0.9, 0, syn)   # negative prior: all-ones decision
    assert orc.words_to_pos(corr, 7) == list(range(7))


def _shor(kats):
    d = kats["shor"]
    return (orc.Code(d["n"], rows=d["hx"]), orc.Code(d["n"], rows=d["hz"]),
            orc.Code(d["n"], rows=d["lx"]), orc.Code(d["n"], rows=d["lz"]), d)


def test_css_syndrome_shor(kats):
    hx, hz, lx, lz, d = _shor(kats)
    c = d["syndrome_case"]
    sx, sz = orc.css_syndrome_words(hx, hz, orc.pos_to_words(c["error_x_part"], 9),
                                    orc.pos_to_words(c["error_z_part"], 9))
    assert orc.words_to_pos(sx, hx.m) == c["syndrome_x"]
    assert orc.words_to_pos(sz, hz.m) == c["syndrome_z"]


def test_css_has_logical_and_stabilizer_shor(kats):
    hx, hz, lx, lz, d = _shor(kats)
    for case in d["has_logical_cases"]["cases"]:
        sx, sz = orc.css_syndrome_words(hx, hz, orc.pos_to_words(case["x_part"], 9),
                                        orc.pos_to_words(case["z_part"], 9))
        assert (not sx.any() and not sz.any()) == case["expected"]
    for case in d["has_stabilizer_cases"]["cases"]:
        f = orc.css_residual_fails(hx, hz, lx, lz, orc.pos_to_words(case["x_part"], 9),
                                   orc.pos_to_words(case["z_part"], 9))
        assert (f == 0) == case["expected"]


def test_hgp_repetition_layout(kats):
    d = kats["hgp_repetition3"]
    hx_rows, hz_rows, nq = orc.hypergraph_product(d["repetition_rows"], 3, d["repetition_rows"], 3)
    assert nq == d["num_qubits"]
    hx, hz = orc.Code(nq, rows=hx_rows), orc.Code(nq, rows=hz_rows)
    # has_logical (css/mod.rs:225-227): both syndromes trivial
    sx, sz = orc.css_syndrome_words(hx, hz, orc.pos_to_words(d["logical_x"], nq), orc.pos_to_words([], nq))
    assert not sx.any() and not sz.any()
    sx, sz = orc.css_syndrome_words(hx, hz, orc.pos_to_words([], nq), orc.pos_to_words(d["logical_z"], nq))
    assert not sx.any() and not sz.any()
    # Hx Hz^T = 0 (css/mod.rs:38-41)
    assert not ((orc.dense(hx_rows, nq).astype(int) @ orc.dense(hz_rows, nq).T.astype(int)) % 2).any()


def test_invalid_matrix_rejected():
    with pytest.raises(ValueError):
        orc.Code(4, rows=[[1, 0]])       # not ascending
    with pytest.raises(ValueError):
        orc.Code(4, rows=[[0, 4]])       # out of range
    with pytest.raises(ValueError):
        orc.Code(4, rows=[[1, 1]])       # duplicate


def test_flip_reference_style_equals_incremental():
    rng = np.random.default_rng(7)
    # irregular random matrices, including degree-0/1/2 bits and empty rows
    for trial in range(60):
        n = int(rng.integers(1, 90))
        m = int(rng.integers(0, 60))
        rows = []
        for _ in range(m):
            k = int(rng.integers(0, min(n, 7) + 1))
            rows.append(sorted(int(x) for x in rng.choice(n, size=k, replace=False)))
        code = orc.Code(n, rows=rows)
        for _ in range(5):
            pos = [int(i) for i in np.nonzero(rng.random(n) < 0.15)[0]]
            out_ref, flips_ref = code.flip_decode(pos)
            w, syn, flips = code.flip_decode_words(orc.pos_to_words(pos, n))
            assert orc.words_to_pos(w, n) == out_ref
            assert flips == flips_ref
            assert orc.words_to_pos(syn, m) == code.syndrome_of(out_ref)


def test_noise_statistics_and_determinism():
    n, p = 1000, 0.02
    tot = 0
    for shot in range(400):
        w = orc.sample_bsc_words(2024, shot, p, n)
        assert (w == orc.sample_bsc_words(2024, shot, p, n)).all()
        assert len(w) == orc.nwords(n)
        assert not (int(w[-1]) >> (n - 64 * (len(w) - 1)))          # no bits beyond n
        tot += len(orc.words_to_pos(w, n))
    mean = tot / (400 * n)
    sigma = (p * (1 - p) / (400 * n)) ** 0.5
    assert abs(mean - p) < 5 * sigma
    assert not orc.sample_bsc_words(1, 0, 0.0, 130).any()
    assert orc.words_to_pos(orc.sample_bsc_words(1, 0, 1.0, 130), 130) == list(range(130))


def test_depolarizing_statistics():
    n, p = 625, 0.05
    cx = cy = cz = 0
    shots = 600
    for shot in range(shots):
        x, z = orc.sample_depolarizing_words(5, shot, p, n)
        xs, zs = set(orc.words_to_pos(x, n)), set(orc.words_to_pos(z, n))
        cy += len(xs & zs)
        cx += len(xs - zs)
        cz += len(zs - xs)
    tot = cx + cy + cz
    sigma = (p * (1 - p) / (shots * n)) ** 0.5
    assert abs(tot / (shots * n) - p) < 5 * sigma
    for c in (cx, cy, cz):
        assert abs(c / tot - 1 / 3) < 5 * (2 / 9 / tot) ** 0.5


def test_gap_table_monotone():
    for p in (1e-9, 0.001, 0.02, 0.5, 0.999, 1.0):
        q = orc.gap_table(p).astype(np.int64)
        assert (np.diff(q[1:]) <= 0).all()
        assert abs(q[1] / 2.0 ** 32 - (1 - p)) < 1e-9
