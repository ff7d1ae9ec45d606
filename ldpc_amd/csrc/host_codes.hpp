// Host-side code construction (setup, not the hot path): sparse GF(2) matrices,
// LinearCode = (H, H^T), the seeded random regular graph sampler, the
// hypergraph product and the logical-operator procedure of the reference.
// Citations are relative to the reference repository.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace ldpc {

// Rows of strictly ascending column indices (SparseBinMat of the un-vendored
// sparse-bin-mat crate; semantics per its call sites).
struct SparseBinMat {
    uint32_t ncols = 0;
    std::vector<uint32_t> ptr{0};  // [nrows + 1]
    std::vector<uint32_t> idx;     // [nnz]

    uint32_t nrows() const { return (uint32_t)ptr.size() - 1; }
    uint32_t nnz() const { return (uint32_t)idx.size(); }
    uint32_t row_len(uint32_t r) const { return ptr[r + 1] - ptr[r]; }
    const uint32_t *row(uint32_t r) const { return idx.data() + ptr[r]; }
    uint32_t max_row_len() const;

    // SparseBinMat::new: returns false (and sets err) on unsorted/duplicate/out-of-range
    static bool from_rows(uint32_t ncols, const std::vector<std::vector<uint32_t>> &rows, SparseBinMat &out,
                          std::string &err);
    static SparseBinMat identity(uint32_t n);
    SparseBinMat transposed() const;                        // rows of the result ascending
    SparseBinMat kron(const SparseBinMat &b) const;         // kron_with
    SparseBinMat hconcat(const SparseBinMat &b) const;      // horizontal_concat_with
    bool product_is_zero(const SparseBinMat &bt) const;     // (this * bt^T... see .cpp) == 0
    std::vector<std::vector<uint32_t>> rows() const;
};

// Basis of the nullspace of `a` over GF(2) as sparse rows (dense bitset
// elimination; host algebra for small codes only).
SparseBinMat nullspace(const SparseBinMat &a);

// Rank over GF(2) (dense bitset elimination; host algebra for small codes only).
uint32_t gf2_rank(const SparseBinMat &a);
// Smallest weight of a non-zero sum of rows of `generators` (LinearCode::minimal_distance,
// linear_code/mod.rs:318-335: every combination of generator rows).  0 = no non-trivial codeword;
// returns false when the matrix has more than 30 rows (2^rows sums).
bool minimal_row_span_weight(const SparseBinMat &generators, uint32_t &out);
// CssErasureDecoder::num_bad_errors (decoders/css_erasure.rs:38-57): rank([S | L]) - rank(S), S / L = the
// syndromes / logical syndromes of the single-qubit errors on the erased positions
uint32_t erasure_num_bad_errors(const SparseBinMat &stabs, const SparseBinMat &logicals, const std::vector<uint32_t> &erasure);

// LinearCode (linear_code/mod.rs:52-56) without the eager generator matrix.
struct LinearCode {
    SparseBinMat h;    // parity_check_matrix
    SparseBinMat ht;   // bit_adjacencies = h.transposed()
    uint32_t len() const { return h.ncols; }
    uint32_t num_checks() const { return h.nrows(); }
};

LinearCode make_code(SparseBinMat h);
LinearCode hamming_code();                 // linear_code/mod.rs:161-167
LinearCode repetition_code(uint32_t n);    // linear_code/mod.rs:139-143

// random.rs:56-76 contract: simple bipartite graph, every bit of degree dv, every
// check of degree dc, checks in label order, neighbours ascending.  Returns
// false iff n*dv != m*dc or no simple graph exists (dc > n, dv > m).
bool random_regular_code(uint32_t n, uint32_t m, uint32_t dv, uint32_t dc, uint64_t seed, LinearCode &out);

struct CssCode {
    LinearCode hx, hz;   // stabilizers.x, stabilizers.z   (css/mod.rs:25-28)
    LinearCode lx, lz;   // logicals.x, logicals.z
};

// css/mod.rs:121-146
SparseBinMat hypergraph_product_x_checks(const LinearCode &a, const LinearCode &b);
SparseBinMat hypergraph_product_z_checks(const LinearCode &a, const LinearCode &b);
// css/logicals.rs:9-85 on generator bases computed by nullspace()
void compute_logicals(const LinearCode &x_code, const LinearCode &z_code, SparseBinMat &lx, SparseBinMat &lz);
// css/mod.rs:35-50; err: 1 = different lengths, 2 = non orthogonal
int css_try_new(const LinearCode &x_code, const LinearCode &z_code, CssCode &out);

// Philox4x32-10 on the host (used by the graph sampler; the device copy lives in
// device_common.hpp)
void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

}  // namespace ldpc
