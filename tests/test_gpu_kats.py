"""The reference's own known-answer tests, run on the MI355X through the C ABI
(single-vector drop-in entry points)."""
import pytest

import ldpc_amd
from ldpc_amd import BpDecoder, CssCode, CssDecoder, FlipDecoder, LinearCode

pytestmark = pytest.mark.gpu


def test_syndrome_doctest(kats):
    d = kats["syndrome_doctest"]
    code = LinearCode.from_parity_check_matrix(d["n"], d["rows"])
    assert code.syndrome_of(d["message"]) == d["syndrome"]
    assert code.has_codeword(d["codeword"]) is True
    assert code.has_codeword(d["not_codeword"]) is False
    with pytest.raises(ldpc_amd.LengthMismatch):
        code.syndrome_of([0], length=8)
    with pytest.raises(ldpc_amd.LdpcError):
        code.syndrome_of([2, 1])


def test_flip_hamming(kats):
    dec = FlipDecoder(LinearCode.hamming_code())
    for case in kats["flip_hamming"]["cases"]:
        assert dec.decode(case["in"]) == case["out"]
    with pytest.raises(ldpc_amd.LengthMismatch):
        dec.decode([0], length=9)


def test_bp_hamming(kats):
    d = kats["bp_hamming"]
    code = LinearCode.hamming_code()
    dec = BpDecoder(code, d["probability"], d["max_iterations"])
    side = kats["session_derived_bp_side_values"]
    for i, case in enumerate(d["cases"]):
        assert dec.decode(case["in"]) == case["out"]
        assert dec.correction_for(code.syndrome_of(case["in"])) == side["corrections"][i]
    with pytest.raises(ldpc_amd.LengthMismatch):
        dec.correction_for([0], length=4)


def test_bp_iteration_cap_zero():
    code = LinearCode.hamming_code()
    syn = code.syndrome_of([0])
    assert BpDecoder(code, 0.1, 0).correction_for(syn) == []
    assert BpDecoder(code, 0.9, 0).correction_for(syn) == list(range(7))


def test_css_shor(kats):
    d = kats["shor"]
    code = CssCode.shor_code()
    c = d["syndrome_case"]
    assert code.syndrome_of(c["error_x_part"], c["error_z_part"]) == (c["syndrome_x"], c["syndrome_z"])
    for case in d["has_logical_cases"]["cases"]:
        assert code.has_logical(case["x_part"], case["z_part"]) is case["expected"]
    for case in d["has_stabilizer_cases"]["cases"]:
        assert code.has_stabilizer(case["x_part"], case["z_part"]) is case["expected"]


def test_hgp_logicals(kats):
    d = kats["hgp_repetition3"]
    rep = LinearCode.repetition_code(3)
    surface = CssCode.hypergraph_product(rep, rep)
    assert surface.has_logical(d["logical_x"], [])
    assert surface.has_logical([], d["logical_z"])
    assert not surface.has_stabilizer(d["logical_x"], [])
    assert not surface.has_stabilizer([], d["logical_z"])


def test_css_decoder_swap_convention(kats):
    # decoders/css.rs:10-15 on the Shor code: correction.z comes from the x decoder
    # (BP on Hx, fed syndrome.x), correction.x from the z decoder (BP on Hz)
    import numpy as np
    from oracle import oracle as orc
    d = kats["shor"]
    code = CssCode.shor_code()
    dec = CssDecoder(code, 0.05, 20)
    hx, hz = orc.Code(9, rows=d["hx"]), orc.Code(9, rows=d["hz"])
    for xp, zp in (([], [7]), ([1], []), ([1], [7]), ([0, 4], [2, 3])):
        sx, sz = code.syndrome_of(xp, zp)
        cx, cz = dec.correction_for(sx, sz)
        ocz, _, _ = hx.bp_correction_for(0.05, 20, orc.pos_to_words(sx, 2))
        ocx, _, _ = hz.bp_correction_for(0.05, 20, orc.pos_to_words(sz, 6))
        assert cx == orc.words_to_pos(ocx, 9) and cz == orc.words_to_pos(ocz, 9)
    # a single X error is unambiguous for the weight-2 Z checks: corrected exactly
    sx, sz = code.syndrome_of([1], [])
    cx, cz = dec.correction_for(sx, sz)
    assert cz == [] and code.syndrome_of(cx, []) == ([], sz)
    assert code.has_stabilizer(sorted(set(cx) ^ {1}), [])
