// Host-side code construction; see host_codes.hpp.  Setup only: nothing here is
// on the Monte-Carlo hot path, which runs entirely in the HIP kernels.
#include "host_codes.hpp"

#include <algorithm>
#include <numeric>

namespace ldpc {

// ---------------------------------------------------------------------------
// SparseBinMat
// ---------------------------------------------------------------------------
uint32_t SparseBinMat::max_row_len() const
{
    uint32_t best = 0;
    for (uint32_t r = 0; r < nrows(); r++) best = std::max(best, row_len(r));
    return best;
}

bool SparseBinMat::from_rows(uint32_t ncols, const std::vector<std::vector<uint32_t>> &rows, SparseBinMat &out,
                             std::string &err)
{
    out = SparseBinMat();
    out.ncols = ncols;
    for (size_t r = 0; r < rows.size(); r++) {
        for (size_t k = 0; k < rows[r].size(); k++) {
            if (rows[r][k] >= ncols) {
                err = "position " + std::to_string(rows[r][k]) + " in row " + std::to_string(r) +
                      " is out of bound for " + std::to_string(ncols) + " columns";
                return false;
            }
            if (k > 0 && rows[r][k] <= rows[r][k - 1]) {
                err = "positions of row " + std::to_string(r) + " are not sorted or contain duplicates";
                return false;
            }
            out.idx.push_back(rows[r][k]);
        }
        out.ptr.push_back((uint32_t)out.idx.size());
    }
    return true;
}

SparseBinMat SparseBinMat::identity(uint32_t n)
{
    SparseBinMat m;
    m.ncols = n;
    m.idx.resize(n);
    std::iota(m.idx.begin(), m.idx.end(), 0u);
    m.ptr.resize(n + 1);
    std::iota(m.ptr.begin(), m.ptr.end(), 0u);
    return m;
}

SparseBinMat SparseBinMat::transposed() const
{
    SparseBinMat t;
    t.ncols = nrows();
    t.ptr.assign((size_t)ncols + 1, 0);
    t.idx.resize(idx.size());
    for (uint32_t c : idx) t.ptr[c + 1]++;
    for (uint32_t c = 0; c < ncols; c++) t.ptr[c + 1] += t.ptr[c];
    std::vector<uint32_t> fill(t.ptr.begin(), t.ptr.end() - 1);
    for (uint32_t r = 0; r < nrows(); r++)   // ascending r => ascending inside each output row
        for (uint32_t e = ptr[r]; e < ptr[r + 1]; e++) t.idx[fill[idx[e]]++] = r;
    return t;
}

SparseBinMat SparseBinMat::kron(const SparseBinMat &b) const
{
    SparseBinMat k;
    k.ncols = ncols * b.ncols;
    k.ptr.clear();
    k.ptr.push_back(0);
    for (uint32_t i = 0; i < nrows(); i++)
        for (uint32_t j = 0; j < b.nrows(); j++) {
            for (uint32_t ea = ptr[i]; ea < ptr[i + 1]; ea++)
                for (uint32_t eb = b.ptr[j]; eb < b.ptr[j + 1]; eb++) k.idx.push_back(idx[ea] * b.ncols + b.idx[eb]);
            k.ptr.push_back((uint32_t)k.idx.size());
        }
    return k;
}

SparseBinMat SparseBinMat::hconcat(const SparseBinMat &b) const
{
    SparseBinMat h;
    h.ncols = ncols + b.ncols;
    h.ptr.clear();
    h.ptr.push_back(0);
    uint32_t rows = std::max(nrows(), b.nrows());
    for (uint32_t r = 0; r < rows; r++) {
        if (r < nrows())
            for (uint32_t e = ptr[r]; e < ptr[r + 1]; e++) h.idx.push_back(idx[e]);
        if (r < b.nrows())
            for (uint32_t e = b.ptr[r]; e < b.ptr[r + 1]; e++) h.idx.push_back(ncols + b.idx[e]);
        h.ptr.push_back((uint32_t)h.idx.size());
    }
    return h;
}

// true iff this * other^T == 0 over GF(2): every row of `this` overlaps every
// row of `other` on an even number of columns (css/mod.rs:38-41).
bool SparseBinMat::product_is_zero(const SparseBinMat &other) const
{
    SparseBinMat ot = other.transposed();   // column -> rows of `other`
    std::vector<uint8_t> acc(other.nrows());
    for (uint32_t r = 0; r < nrows(); r++) {
        std::fill(acc.begin(), acc.end(), 0);
        for (uint32_t e = ptr[r]; e < ptr[r + 1]; e++) {
            uint32_t c = idx[e];
            for (uint32_t k = ot.ptr[c]; k < ot.ptr[c + 1]; k++) acc[ot.idx[k]] ^= 1;
        }
        for (uint8_t a : acc)
            if (a) return false;
    }
    return true;
}

std::vector<std::vector<uint32_t>> SparseBinMat::rows() const
{
    std::vector<std::vector<uint32_t>> out(nrows());
    for (uint32_t r = 0; r < nrows(); r++) out[r].assign(idx.begin() + ptr[r], idx.begin() + ptr[r + 1]);
    return out;
}

// ---------------------------------------------------------------------------
// GF(2) dense helpers
// ---------------------------------------------------------------------------
namespace {
struct BitRows {
    uint32_t ncols, words;
    std::vector<std::vector<uint64_t>> r;
    explicit BitRows(uint32_t nc) : ncols(nc), words((nc + 63) / 64) {}
    void push(const SparseBinMat &m, uint32_t row)
    {
        std::vector<uint64_t> v(words, 0);
        for (uint32_t e = m.ptr[row]; e < m.ptr[row + 1]; e++) v[m.idx[e] >> 6] |= 1ull << (m.idx[e] & 63);
        r.push_back(std::move(v));
    }
};
inline bool bit(const std::vector<uint64_t> &v, uint32_t i) { return (v[i >> 6] >> (i & 63)) & 1; }
inline void xor_into(std::vector<uint64_t> &a, const std::vector<uint64_t> &b)
{
    for (size_t w = 0; w < a.size(); w++) a[w] ^= b[w];
}
inline bool odd_overlap(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b)
{
    unsigned p = 0;
    for (size_t w = 0; w < a.size(); w++) p ^= (unsigned)__builtin_popcountll(a[w] & b[w]);
    return p & 1;
}
std::vector<uint32_t> positions(const std::vector<uint64_t> &v, uint32_t n)
{
    std::vector<uint32_t> p;
    for (uint32_t i = 0; i < n; i++)
        if (bit(v, i)) p.push_back(i);
    return p;
}
}  // namespace

SparseBinMat nullspace(const SparseBinMat &a)
{
    uint32_t n = a.ncols;
    BitRows m(n);
    for (uint32_t r = 0; r < a.nrows(); r++) m.push(a, r);
    std::vector<int64_t> pivot_row_of_col(n, -1);
    uint32_t rank = 0;
    for (uint32_t c = 0; c < n && rank < m.r.size(); c++) {
        size_t piv = rank;
        while (piv < m.r.size() && !bit(m.r[piv], c)) piv++;
        if (piv == m.r.size()) continue;
        std::swap(m.r[rank], m.r[piv]);
        for (size_t i = 0; i < m.r.size(); i++)
            if (i != rank && bit(m.r[i], c)) xor_into(m.r[i], m.r[rank]);
        pivot_row_of_col[c] = rank++;
    }
    std::vector<std::vector<uint32_t>> basis;
    for (uint32_t f = 0; f < n; f++) {
        if (pivot_row_of_col[f] >= 0) continue;
        std::vector<uint32_t> v;
        for (uint32_t c = 0; c < n; c++) {
            if (c == f) v.push_back(c);
            else if (pivot_row_of_col[c] >= 0 && bit(m.r[(size_t)pivot_row_of_col[c]], f)) v.push_back(c);
        }
        basis.push_back(std::move(v));
    }
    SparseBinMat out;
    std::string err;
    SparseBinMat::from_rows(n, basis, out, err);
    return out;
}

static uint32_t rank_of(std::vector<std::vector<uint64_t>> &rows, uint32_t ncols)
{
    uint32_t rank = 0;
    for (uint32_t c = 0; c < ncols && rank < rows.size(); c++) {
        size_t piv = rank;
        while (piv < rows.size() && !bit(rows[piv], c)) piv++;
        if (piv == rows.size()) continue;
        std::swap(rows[rank], rows[piv]);
        for (size_t i = rank + 1; i < rows.size(); i++)
            if (bit(rows[i], c)) xor_into(rows[i], rows[rank]);
        rank++;
    }
    return rank;
}

uint32_t gf2_rank(const SparseBinMat &a)
{
    BitRows m(a.ncols);
    for (uint32_t r = 0; r < a.nrows(); r++) m.push(a, r);
    return rank_of(m.r, a.ncols);
}

bool minimal_row_span_weight(const SparseBinMat &generators, uint32_t &out)
{
    const uint32_t k = generators.nrows();
    if (k > 30) return false;
    BitRows m(generators.ncols);
    for (uint32_t r = 0; r < k; r++) m.push(generators, r);
    std::vector<uint64_t> sum(m.words, 0);
    uint32_t best = 0;
    // Gray code: consecutive sums differ by one generator; every non-empty subset is visited once
    for (uint64_t i = 1; i < (1ull << k); i++) {
        xor_into(sum, m.r[(size_t)__builtin_ctzll(i)]);
        uint32_t w = 0;
        for (uint64_t x : sum) w += (uint32_t)__builtin_popcountll(x);
        if (w > 0 && (best == 0 || w < best)) best = w;
    }
    out = best;
    return true;
}

uint32_t erasure_num_bad_errors(const SparseBinMat &stabs, const SparseBinMat &logicals, const std::vector<uint32_t> &erasure)
{
    // row e of S: the stabilizers that contain erased qubit e (its syndrome); likewise L; T = [S | L]
    const SparseBinMat st = stabs.transposed(), lt = logicals.transposed();
    const uint32_t ns = stabs.nrows(), nl = logicals.nrows();
    std::vector<std::vector<uint64_t>> S, T;
    for (uint32_t q : erasure) {
        std::vector<uint64_t> s((ns + 63) / 64, 0), t((ns + nl + 63) / 64, 0);
        for (uint32_t e = st.ptr[q]; e < st.ptr[q + 1]; e++) {
            s[st.idx[e] >> 6] |= 1ull << (st.idx[e] & 63);
            t[st.idx[e] >> 6] |= 1ull << (st.idx[e] & 63);
        }
        for (uint32_t e = lt.ptr[q]; e < lt.ptr[q + 1]; e++) {
            const uint32_t c = ns + lt.idx[e];
            t[c >> 6] |= 1ull << (c & 63);
        }
        S.push_back(std::move(s));
        T.push_back(std::move(t));
    }
    return rank_of(T, ns + nl) - rank_of(S, ns);
}

// ---------------------------------------------------------------------------
// LinearCode
// ---------------------------------------------------------------------------
LinearCode make_code(SparseBinMat h)
{
    LinearCode c;
    c.ht = h.transposed();
    c.h = std::move(h);
    return c;
}

LinearCode hamming_code()
{
    SparseBinMat h;
    std::string err;
    SparseBinMat::from_rows(7, {{3, 4, 5, 6}, {1, 2, 5, 6}, {0, 2, 4, 6}}, h, err);
    return make_code(std::move(h));
}

LinearCode repetition_code(uint32_t n)
{
    std::vector<std::vector<uint32_t>> rows;
    for (uint32_t c = 0; c + 1 < n; c++) rows.push_back({c, c + 1});
    SparseBinMat h;
    std::string err;
    SparseBinMat::from_rows(n, rows, h, err);
    return make_code(std::move(h));
}

// ---------------------------------------------------------------------------
// Philox4x32-10 and the seeded regular-graph sampler
// ---------------------------------------------------------------------------
void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

namespace {
struct HostRng {
    uint32_t key[2], ctr[4] = {0, 0, 0, 0x67726170u /* "grap": graph-sampler stream */}, buf[4];
    int have = 0;
    explicit HostRng(uint64_t seed) { key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32); }
    uint32_t next()
    {
        if (!have) {
            philox4x32_10(ctr, key, buf);
            if (++ctr[0] == 0) ++ctr[1];
            have = 4;
        }
        return buf[4 - have--];
    }
    uint32_t below(uint32_t k) { return (uint32_t)(((uint64_t)next() * k) >> 32); }
};

bool row_has(const std::vector<uint32_t> &sock, size_t base, uint32_t dc, uint32_t bit, size_t skip)
{
    for (size_t j = base; j < base + dc; j++)
        if (j != skip && sock[j] == bit) return true;
    return false;
}
}  // namespace

bool random_regular_code(uint32_t n, uint32_t m, uint32_t dv, uint32_t dc, uint64_t seed, LinearCode &out)
{
    if ((uint64_t)n * dv != (uint64_t)m * dc) return false;
    if (dc > n || dv > m) return false;
    uint64_t E64 = (uint64_t)n * dv;
    if (E64 >= 0xFFFFFFFFull) return false;
    size_t E = (size_t)E64;
    HostRng rng(seed);
    std::vector<uint32_t> sock(E);
    // configuration model: bit b owns dv sockets, a random permutation deals
    // them to the checks, dc per check in label order
    for (size_t s = 0; s < E; s++) sock[s] = (uint32_t)(s / dv);
    for (size_t s = E; s > 1; s--) std::swap(sock[s - 1], sock[rng.below((uint32_t)s)]);
    // repair multi-edges by switching with random sockets that keep both rows simple
    bool stuck = false;
    for (size_t s = 0; s < E && !stuck; s++) {
        size_t base = s - s % dc;
        if (!row_has(sock, base, dc, sock[s], s)) continue;
        int tries = 0;
        for (; tries < 4096; tries++) {
            size_t t = rng.below((uint32_t)E), tb = t - t % dc;
            if (tb == base) continue;
            if (row_has(sock, base, dc, sock[t], s) || row_has(sock, tb, dc, sock[s], t)) continue;
            std::swap(sock[s], sock[t]);
            break;
        }
        if (tries == 4096) stuck = true;
    }
    if (stuck) {
        // dense corner (e.g. the complete graph of examples/random_linear_code.rs):
        // start from a circulant simple regular graph and randomise it with
        // double-edge switches that preserve simplicity
        for (size_t s = 0; s < E; s++) sock[s] = (uint32_t)(s % n);
        for (size_t it = 0; it < 20 * E; it++) {
            size_t s = rng.below((uint32_t)E), t = rng.below((uint32_t)E);
            size_t sb = s - s % dc, tb = t - t % dc;
            if (sb == tb) continue;
            if (row_has(sock, sb, dc, sock[t], s) || row_has(sock, tb, dc, sock[s], t)) continue;
            std::swap(sock[s], sock[t]);
        }
    }
    std::vector<std::vector<uint32_t>> rows(m);
    for (uint32_t c = 0; c < m; c++) {
        rows[c].assign(sock.begin() + (size_t)c * dc, sock.begin() + (size_t)(c + 1) * dc);
        std::sort(rows[c].begin(), rows[c].end());   // random.rs:72 neighbours ascending
    }
    SparseBinMat h;
    std::string err;
    if (!SparseBinMat::from_rows(n, rows, h, err)) return false;
    out = make_code(std::move(h));
    return true;
}

// ---------------------------------------------------------------------------
// CSS
// ---------------------------------------------------------------------------
SparseBinMat hypergraph_product_x_checks(const LinearCode &a, const LinearCode &b)
{
    // [ I_{n_a} (x) H_b | H_a^T (x) I_{m_b} ]   css/mod.rs:121-133
    return SparseBinMat::identity(a.len()).kron(b.h).hconcat(a.ht.kron(SparseBinMat::identity(b.num_checks())));
}

SparseBinMat hypergraph_product_z_checks(const LinearCode &a, const LinearCode &b)
{
    // [ H_a (x) I_{n_b} | I_{m_a} (x) H_b^T ]   css/mod.rs:135-146
    return a.h.kron(SparseBinMat::identity(b.len())).hconcat(SparseBinMat::identity(a.num_checks()).kron(b.ht));
}

void compute_logicals(const LinearCode &x_code, const LinearCode &z_code, SparseBinMat &lx, SparseBinMat &lz)
{
    uint32_t n = x_code.len();
    // logicals.rs:22-30: raw X generators come from the Z code's generator
    // matrix (codewords of ker Hz), raw Z generators from the X code's
    BitRows raw_x(n), raw_z(n);
    SparseBinMat gz = nullspace(z_code.h), gx = nullspace(x_code.h);
    for (uint32_t r = 0; r < gz.nrows(); r++) raw_x.push(gz, r);
    for (uint32_t r = 0; r < gx.nrows(); r++) raw_z.push(gx, r);
    std::vector<std::vector<uint32_t>> xl, zl;
    while (!raw_x.r.empty()) {                                   // logicals.rs:40
        std::vector<uint64_t> xg = std::move(raw_x.r.back());
        raw_x.r.pop_back();
        size_t pos = 0;                                          // :56-64 first anticommuting partner
        while (pos < raw_z.r.size() && !odd_overlap(xg, raw_z.r[pos])) pos++;
        if (pos == raw_z.r.size()) continue;
        std::vector<uint64_t> zg = std::move(raw_z.r[pos]);
        raw_z.r[pos] = std::move(raw_z.r.back());                // swap_remove
        raw_z.r.pop_back();
        for (auto &g : raw_z.r)                                  // :71-74
            if (odd_overlap(xg, g)) xor_into(g, zg);
        for (auto &g : raw_x.r)                                  // :75-78
            if (odd_overlap(g, zg)) xor_into(g, xg);
        xl.push_back(positions(xg, n));                          // :81-84
        zl.push_back(positions(zg, n));
    }
    std::string err;
    SparseBinMat::from_rows(n, xl, lx, err);
    SparseBinMat::from_rows(n, zl, lz, err);
}

int css_try_new(const LinearCode &x_code, const LinearCode &z_code, CssCode &out)
{
    if (x_code.len() != z_code.len()) return 1;                  // css/mod.rs:36-37
    if (!x_code.h.product_is_zero(z_code.h)) return 2;           // :38-41
    out.hx = x_code;
    out.hz = z_code;
    SparseBinMat lx, lz;
    compute_logicals(x_code, z_code, lx, lz);                    // :48
    out.lx = make_code(std::move(lx));
    out.lz = make_code(std::move(lz));
    return 0;
}

}  // namespace ldpc
