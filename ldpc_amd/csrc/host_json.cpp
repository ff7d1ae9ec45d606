// See host_json.hpp.  A small recursive-descent reader for the subset of JSON these documents use (objects,
// arrays of non-negative integers) that skips anything else it meets inside an object.
#include "host_json.hpp"

#include <cstring>

namespace ldpc {

namespace {

void append_list(std::string &s, const std::vector<uint32_t> &v)
{
    s += '[';
    char buf[16];
    for (size_t i = 0; i < v.size(); i++) {
        if (i) s += ',';
        int k = 0;
        uint32_t x = v[i];
        do { buf[k++] = (char)('0' + x % 10); x /= 10; } while (x);
        while (k) s += buf[--k];
    }
    s += ']';
}

struct Reader {
    const char *p, *end;
    std::string err;

    bool fail(const std::string &what)
    {
        if (err.empty()) err = "json: " + what + " at byte " + std::to_string((size_t)(p - begin));
        return false;
    }
    const char *begin;

    void ws()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
    }
    bool eat(char c)
    {
        ws();
        if (p < end && *p == c) { p++; return true; }
        return false;
    }
    bool expect(char c)
    {
        if (eat(c)) return true;
        return fail(std::string("expected '") + c + "'");
    }
    bool string(std::string &out)
    {
        ws();
        if (p >= end || *p != '"') return fail("expected a string");
        p++;
        out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\') {               // member names here hold no escapes; keep the escaped byte as it is
                p++;
                if (p >= end) break;
            }
            out += *p++;
        }
        if (p >= end) return fail("unterminated string");
        p++;
        return true;
    }
    bool number(uint64_t &out)
    {
        ws();
        if (p >= end || *p < '0' || *p > '9') return fail("expected a non-negative integer");
        out = 0;
        while (p < end && *p >= '0' && *p <= '9') {
            if (out > (UINT64_MAX - 9) / 10) return fail("integer too large");
            out = out * 10 + (uint64_t)(*p++ - '0');
        }
        if (p < end && (*p == '.' || *p == 'e' || *p == 'E')) return fail("expected an integer");
        return true;
    }
    bool list(std::vector<uint64_t> &out)
    {
        out.clear();
        if (!expect('[')) return false;
        if (eat(']')) return true;
        do {
            uint64_t v;
            if (!number(v)) return false;
            out.push_back(v);
        } while (eat(','));
        return expect(']');
    }
    // any value, discarded
    bool skip(int depth = 0)
    {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        if (*p == '{') {
            p++;
            if (eat('}')) return true;
            do {
                std::string k;
                if (!string(k) || !expect(':') || !skip(depth + 1)) return false;
            } while (eat(','));
            return expect('}');
        }
        if (*p == '[') {
            p++;
            if (eat(']')) return true;
            do {
                if (!skip(depth + 1)) return false;
            } while (eat(','));
            return expect(']');
        }
        if (*p == '"') {
            std::string s;
            return string(s);
        }
        const char *q = p;   // number / true / false / null
        while (p < end && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n' && *p != '\t' && *p != '\r') p++;
        return p > q ? true : fail("unexpected character");
    }

    // calls member(name) for every member of an object; member consumes the value
    template <class F>
    bool object(F member)
    {
        if (!expect('{')) return false;
        if (eat('}')) return true;
        do {
            std::string k;
            if (!string(k) || !expect(':')) return false;
            if (!member(k)) return false;
        } while (eat(','));
        return expect('}');
    }

    bool matrix(SparseBinMat &out)
    {
        std::vector<uint64_t> ranges, cols;
        uint64_t ncols = 0;
        bool have_r = false, have_c = false, have_n = false;
        bool ok = object([&](const std::string &k) {
            if (k == "row_ranges") return have_r = list(ranges);
            if (k == "column_indices") return have_c = list(cols);
            if (k == "number_of_columns") return have_n = number(ncols);
            return skip();
        });
        if (!ok) return false;
        if (!have_r || !have_c || !have_n) return fail("matrix needs row_ranges, column_indices and number_of_columns");
        if (ncols >= 0xFFFFFFFFull || cols.size() >= 0xFFFFFFFFull || ranges.size() >= 0xFFFFFFFFull)
            return fail("matrix dimension exceeds 32 bits");
        if (ranges.empty()) ranges.push_back(0);   // a matrix without rows
        if (ranges[0] != 0 || ranges.back() != cols.size()) return fail("row_ranges do not cover column_indices");
        std::vector<std::vector<uint32_t>> rows(ranges.size() - 1);
        for (size_t r = 0; r + 1 < ranges.size(); r++) {
            if (ranges[r + 1] < ranges[r] || ranges[r + 1] > cols.size()) return fail("row_ranges must be non-decreasing");
            for (uint64_t e = ranges[r]; e < ranges[r + 1]; e++) {
                if (cols[e] >= ncols) return fail("column index out of bound");
                rows[r].push_back((uint32_t)cols[e]);
            }
        }
        std::string e;
        if (!SparseBinMat::from_rows((uint32_t)ncols, rows, out, e)) return fail(e);
        return true;
    }

    bool css_pair(SparseBinMat &x, SparseBinMat &z)
    {
        bool hx = false, hz = false;
        bool ok = object([&](const std::string &k) {
            if (k == "x") return hx = matrix(x);
            if (k == "z") return hz = matrix(z);
            return skip();
        });
        if (!ok) return false;
        return hx && hz ? true : fail("expected members x and z");
    }

    bool done()
    {
        ws();
        return p == end ? true : fail("trailing characters");
    }
};

bool same(const SparseBinMat &a, const SparseBinMat &b) { return a.ncols == b.ncols && a.ptr == b.ptr && a.idx == b.idx; }

}  // namespace

std::string sparse_bin_mat_to_json(const SparseBinMat &m)
{
    std::string s;
    s.reserve(64 + 8 * ((size_t)m.nnz() + m.ptr.size()));
    s += "{\"row_ranges\":";
    append_list(s, m.ptr);
    s += ",\"column_indices\":";
    append_list(s, m.idx);
    s += ",\"number_of_columns\":" + std::to_string(m.ncols) + "}";
    return s;
}

std::string linear_code_to_json(const LinearCode &c)
{
    return "{\"parity_check_matrix\":" + sparse_bin_mat_to_json(c.h) + ",\"generator_matrix\":" +
           sparse_bin_mat_to_json(nullspace(c.h)) + ",\"bit_adjacencies\":" + sparse_bin_mat_to_json(c.ht) + "}";
}

std::string css_code_to_json(const CssCode &c)
{
    return "{\"stabilizers\":{\"x\":" + sparse_bin_mat_to_json(c.hx.h) + ",\"z\":" + sparse_bin_mat_to_json(c.hz.h) +
           "},\"logicals\":{\"x\":" + sparse_bin_mat_to_json(c.lx.h) + ",\"z\":" + sparse_bin_mat_to_json(c.lz.h) + "}}";
}

bool linear_code_from_json(const char *text, size_t len, LinearCode &out, std::string &err)
{
    Reader r{text, text + len, {}, text};
    SparseBinMat h, g, ht;
    bool have_h = false, have_g = false, have_t = false;
    bool ok = r.object([&](const std::string &k) {
        if (k == "parity_check_matrix") return have_h = r.matrix(h);
        if (k == "generator_matrix") return have_g = r.matrix(g);
        if (k == "bit_adjacencies") return have_t = r.matrix(ht);
        return r.skip();
    });
    ok = ok && r.done();
    if (ok && !have_h) ok = r.fail("expected member parity_check_matrix");
    if (!ok) {
        err = r.err;
        return false;
    }
    LinearCode c = make_code(std::move(h));
    if (have_t && !same(ht, c.ht)) {
        err = "json: bit_adjacencies is not the transpose of parity_check_matrix";
        return false;
    }
    if (have_g && (g.ncols != c.h.ncols || !c.h.product_is_zero(g))) {   // from_both_matrices, linear_code/mod.rs:59-72
        err = g.ncols != c.h.ncols ? "json: matrics have different number of bits (columns)" : "json: matrices are non orthogonal";
        return false;
    }
    out = std::move(c);
    return true;
}

bool css_code_from_json(const char *text, size_t len, CssCode &out, std::string &err)
{
    Reader r{text, text + len, {}, text};
    SparseBinMat hx, hz, lx, lz;
    bool have_s = false, have_l = false;
    bool ok = r.object([&](const std::string &k) {
        if (k == "stabilizers") return have_s = r.css_pair(hx, hz);
        if (k == "logicals") return have_l = r.css_pair(lx, lz);
        return r.skip();
    });
    ok = ok && r.done();
    if (ok && !(have_s && have_l)) ok = r.fail("expected members stabilizers and logicals");
    if (!ok) {
        err = r.err;
        return false;
    }
    if (hz.ncols != hx.ncols || lx.ncols != hx.ncols || lz.ncols != hx.ncols) {
        err = "json: different x and z lengths";
        return false;
    }
    if (!hx.product_is_zero(hz)) {   // css/mod.rs:38-42
        err = "json: codes are not orthogonal";
        return false;
    }
    out.hx = make_code(std::move(hx));
    out.hz = make_code(std::move(hz));
    out.lx = make_code(std::move(lx));
    out.lz = make_code(std::move(lz));
    return true;
}

}  // namespace ldpc
