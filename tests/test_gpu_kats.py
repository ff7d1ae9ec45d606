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


def test_css_decoder_swap_convention():
    # decoders/css.rs:10-15 on the Shor code: a single Z error on qubit 7 fires
    # x-stabilizer 1; the correction must come back in the z part
    code = CssCode.shor_code()
    dec = CssDecoder(code, 0.05, 20)
    sx, sz = code.syndrome_of([], [7])
    cx, cz = dec.correction_for(sx, sz)
    assert cx == []
    assert code.has_stabilizer([], sorted(set(cz) ^ {7}))
    sx, sz = code.syndrome_of([1], [])
    cx, cz = dec.correction_for(sx, sz)
    assert cz == []
    assert code.syndrome_of(cx, []) == ([], sz)
