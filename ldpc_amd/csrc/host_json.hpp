// JSON text of codes, in the layout the reference's serde derives would emit
// (LinearCode: linear_code/mod.rs:51-56, as_json :468-471; CssCode: css/mod.rs:24-28; Css<T>: css.rs).
//
// PARITY UNPINNED: the innermost object, SparseBinMat, belongs to the un-vendored sparse-bin-mat 0.7 crate.  Its
// fields are taken to be {row_ranges, column_indices, number_of_columns} (a derive on the struct's private fields);
// no file of the reference holds a sample, so this is an assumption until a box with the crate exports one.
// Host-side set-up code, not on the decoding path.
#pragma once
#include <string>

#include "host_codes.hpp"

namespace ldpc {

std::string sparse_bin_mat_to_json(const SparseBinMat &m);
// {"parity_check_matrix":M,"generator_matrix":M,"bit_adjacencies":M}; the generator matrix is nullspace(h)
std::string linear_code_to_json(const LinearCode &c);
// {"stabilizers":{"x":M,"z":M},"logicals":{"x":M,"z":M}}
std::string css_code_to_json(const CssCode &c);

// Parsers: false + err on malformed text, missing members or malformed matrices.  Members are accepted in any
// order; unknown members are skipped.  linear_code_from_json needs parity_check_matrix; when bit_adjacencies /
// generator_matrix are present they must be its transpose / orthogonal to it with the same number of columns.
bool linear_code_from_json(const char *text, size_t len, LinearCode &out, std::string &err);
bool css_code_from_json(const char *text, size_t len, CssCode &out, std::string &err);

}  // namespace ldpc
