#include <cstdio>
#include <random>
#include "host_codes.hpp"
using namespace ldpc;
int main() {
    std::mt19937 rng(11);
    for (int it = 0; it < 3000; it++) {
        uint32_t n = rng() % 70, m = rng() % 50;
        std::vector<std::vector<uint32_t>> rows(m);
        for (auto &r : rows) for (uint32_t c = 0; c < n; c++) if (rng() % 5 == 0) r.push_back(c);
        SparseBinMat h; std::string err;
        if (!SparseBinMat::from_rows(n, rows, h, err)) { printf("from_rows failed\n"); return 1; }
        LinearCode code = make_code(h);
        SparseBinMat g = nullspace(h);
        if (!h.product_is_zero(g)) { printf("nullspace not orthogonal\n"); return 1; }
        if (g.nrows() + gf2_rank(h) != n) { printf("rank-nullity broken n=%u\n", n); return 1; }
        uint32_t w = 0;
        if (g.nrows() <= 12) minimal_row_span_weight(g, w);
        if (it % 50 == 0 && n >= 2 && n <= 12) {
            CssCode c;
            LinearCode rep = repetition_code(n);
            SparseBinMat hx = hypergraph_product_x_checks(rep, rep), hz = hypergraph_product_z_checks(rep, rep);
            if (css_try_new(make_code(hx), make_code(hz), c) != 0) { printf("hgp not orthogonal\n"); return 1; }
            std::vector<uint32_t> er;
            for (uint32_t q = 0; q < c.hx.len(); q++) if (rng() % 3 == 0) er.push_back(q);
            erasure_num_bad_errors(c.hx.h, c.lx.h, er);
        }
    }
    for (int it = 0; it < 200; it++) {
        uint32_t dv = 2 + rng() % 4, dc = dv + 1 + rng() % 4, k = 1 + rng() % 30;
        LinearCode c;
        bool ok = random_regular_code(dc * k, dv * k, dv, dc, rng(), c);
        if (ok) {
            if (c.h.nnz() != dc * k * dv) { printf("edge count\n"); return 1; }
            if (c.h.max_row_len() != dc || c.ht.max_row_len() != dv) { printf("degrees\n"); return 1; }
        }
    }
    printf("host algebra clean\n");
    return 0;
}
