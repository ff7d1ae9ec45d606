"""JSON text of codes: LinearCode::as_json, linear_code/mod.rs:468-471, and the serde
derives of LinearCode (mod.rs:51-56) and CssCode (css/mod.rs:24-28, css.rs:7-11).

PARITY UNPINNED: the reference holds no sample of this text and the layout of the innermost object belongs to
the un-vendored sparse-bin-mat crate; what is tested is the documented assumption
({"row_ranges", "column_indices", "number_of_columns"}), round trips, and the checks on import.  Host code only."""
import json

import pytest

import ldpc_amd as L

HAMMING_ROWS = [[3, 4, 5, 6], [1, 2, 5, 6], [0, 2, 4, 6]]      # linear_code/mod.rs:162-165


def _mat(rows, ncols):
    ranges, cols = [0], []
    for r in rows:
        cols += list(r)
        ranges.append(len(cols))
    return {"row_ranges": ranges, "column_indices": cols, "number_of_columns": ncols}


def _rows(mat):
    r, c = mat["row_ranges"], mat["column_indices"]
    return [c[r[i]:r[i + 1]] for i in range(len(r) - 1)]


def test_linear_code_text_has_the_three_members_of_the_struct_in_order():
    doc = json.loads(L.LinearCode.hamming_code().as_json())
    assert list(doc) == ["parity_check_matrix", "generator_matrix", "bit_adjacencies"]
    for m in doc.values():
        assert list(m) == ["row_ranges", "column_indices", "number_of_columns"]
    assert doc["parity_check_matrix"] == _mat(HAMMING_ROWS, 7)
    assert _rows(doc["bit_adjacencies"]) == [[2], [1], [1, 2], [0], [0, 2], [0, 1], [0, 1, 2]]
    assert doc["bit_adjacencies"]["number_of_columns"] == 3
    gen = _rows(doc["generator_matrix"])
    assert len(gen) == 4 and doc["generator_matrix"]["number_of_columns"] == 7
    for g in gen:                                   # every generator is a codeword
        assert all(len(set(g) & set(h)) % 2 == 0 for h in HAMMING_ROWS)


def test_linear_code_round_trip():
    for code in (L.LinearCode.hamming_code(), L.LinearCode.repetition_code(5),
                 L.LinearCode.random_regular_code(20, 15, 3, 4, seed=7), L.LinearCode.from_parity_check_matrix(4, [[], [0, 3]])):
        back = L.LinearCode.from_json(code.as_json())
        assert back.parity_check_matrix() == code.parity_check_matrix()
        assert back.len() == code.len() and back.num_checks() == code.num_checks()
        assert back.as_json() == code.as_json()


def test_import_needs_only_the_parity_checks_and_skips_what_it_does_not_know():
    text = json.dumps({"comment": {"nested": [1, {"a": None}], "s": "x\\\"y"}, "parity_check_matrix": _mat(HAMMING_ROWS, 7),
                       "extra": -1.5e3}, indent=2)
    code = L.LinearCode.from_json(text)
    assert code.parity_check_matrix() == HAMMING_ROWS


def test_import_accepts_another_basis_of_the_codespace():
    doc = json.loads(L.LinearCode.hamming_code().as_json())
    doc["generator_matrix"] = _mat([[0, 1, 2]], 7)                # a codeword (flip.rs:60-98 decodes to it), rank 1 only
    assert L.LinearCode.from_json(json.dumps(doc)).len() == 7


@pytest.mark.parametrize("mutate,needle", [
    (lambda d: d.pop("parity_check_matrix"), "parity_check_matrix"),
    (lambda d: d["parity_check_matrix"].pop("row_ranges"), "row_ranges"),
    (lambda d: d["parity_check_matrix"]["column_indices"].__setitem__(0, 7), "out of bound"),
    (lambda d: d["parity_check_matrix"]["column_indices"].__setitem__(1, 3), ""),            # duplicate in a row
    (lambda d: d["parity_check_matrix"]["row_ranges"].__setitem__(3, 11), "row_ranges"),
    (lambda d: d["parity_check_matrix"]["row_ranges"].__setitem__(1, 9), "row_ranges"),
    (lambda d: d["bit_adjacencies"]["column_indices"].__setitem__(0, 1), "transpose"),
    (lambda d: d["generator_matrix"].update(_mat([[0]], 7)), "orthogonal"),
    (lambda d: d["generator_matrix"].update(_mat([[0, 1]], 3)), "columns"),
])
def test_import_rejects_inconsistent_documents(mutate, needle):
    doc = json.loads(L.LinearCode.hamming_code().as_json())
    mutate(doc)
    with pytest.raises(L.LdpcError) as e:
        L.LinearCode.from_json(json.dumps(doc))
    assert needle in str(e.value)


@pytest.mark.parametrize("text", ["", "[]", "{", '{"parity_check_matrix":', '{"parity_check_matrix":{"row_ranges":[0,1.5]}}',
                                  '{"parity_check_matrix":{"row_ranges":[0],"column_indices":[],"number_of_columns":-1}}',
                                  '{"parity_check_matrix":{"row_ranges":[0],"column_indices":[],"number_of_columns":3}} x',
                                  '{"parity_check_matrix":{"row_ranges":[0],"column_indices":[],"number_of_columns":99999999999999999999999}}'])
def test_import_rejects_malformed_text(text):
    with pytest.raises(L.LdpcError):
        L.LinearCode.from_json(text)


def test_css_code_text_and_round_trip():
    shor = L.CssCode.shor_code()
    doc = json.loads(shor.as_json())
    assert list(doc) == ["stabilizers", "logicals"] and list(doc["stabilizers"]) == ["x", "z"]
    assert _rows(doc["stabilizers"]["x"]) == [[0, 1, 2, 3, 4, 5], [3, 4, 5, 6, 7, 8]]        # css/mod.rs:63-66
    assert _rows(doc["stabilizers"]["z"]) == [[0, 1], [1, 2], [3, 4], [4, 5], [6, 7], [7, 8]]
    assert _rows(doc["logicals"]["x"]) == [[0, 1, 2]] and _rows(doc["logicals"]["z"]) == [[0, 3, 6]]
    for css in (shor, L.CssCode.steane_code(), L.CssCode.toric_code(3),
                L.CssCode.hypergraph_product(L.LinearCode.repetition_code(3), L.LinearCode.repetition_code(3))):
        back = L.CssCode.from_json(css.as_json())
        assert back.as_json() == css.as_json()
        assert back.len() == css.len()


def test_css_import_checks_lengths_and_orthogonality():
    doc = json.loads(L.CssCode.shor_code().as_json())
    bad = json.loads(json.dumps(doc))
    bad["stabilizers"]["z"] = _mat([[0]], 9)                      # anticommutes with the first X stabilizer
    with pytest.raises(L.LdpcError) as e:
        L.CssCode.from_json(json.dumps(bad))
    assert "orthogonal" in str(e.value) and e.value.status == L.ERR_CSS
    bad = json.loads(json.dumps(doc))
    bad["logicals"]["x"] = _mat([[0, 1, 2]], 8)
    with pytest.raises(L.LdpcError) as e:
        L.CssCode.from_json(json.dumps(bad))
    assert "lengths" in str(e.value)
    del doc["logicals"]
    with pytest.raises(L.LdpcError):
        L.CssCode.from_json(json.dumps(doc))


@pytest.mark.gpu
def test_imported_codes_decode_like_the_originals():
    """a code that went through its JSON text runs the hot path as the original does"""
    code = L.LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)
    back = L.LinearCode.from_json(code.as_json())
    assert back.syndrome_of([0, 2, 4]) == code.syndrome_of([0, 2, 4])
    a = L.FlipDecoder(code).mc_run(0.02, 2024, 0, 4096)
    b = L.FlipDecoder(back).mc_run(0.02, 2024, 0, 4096)
    assert list(a) == list(b)
    shor = L.CssCode.from_json(L.CssCode.shor_code().as_json())
    assert shor.has_logical(x_part=[], z_part=[0, 3, 6])                                      # css/mod.rs:219-223
    assert not shor.has_logical(x_part=[3], z_part=[0, 6])
    hgp = L.CssCode.hypergraph_product(L.LinearCode.random_regular_code(20, 15, 3, 4, seed=1),
                                       L.LinearCode.random_regular_code(20, 15, 3, 4, seed=1))
    a = L.CssDecoder(hgp, 0.01, 100).mc_run(0.01, 2024, 0, 4096)
    b = L.CssDecoder(L.CssCode.from_json(hgp.as_json()), 0.01, 100).mc_run(0.01, 2024, 0, 4096)
    assert list(a) == list(b)
