#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include "host_json.hpp"
using namespace ldpc;
int main() {
    LinearCode ham = hamming_code();
    std::string base = linear_code_to_json(ham);
    CssCode c; css_try_new(ham, ham, c);
    std::string cbase = css_code_to_json(c);
    std::mt19937 rng(7);
    const char alphabet[] = "{}[]:,\"0123456789 -.eE\\ntruefalsnl";
    size_t ok = 0, bad = 0;
    for (int it = 0; it < 300000; it++) {
        std::string t = (it & 1) ? base : cbase;
        int nm = 1 + rng() % 4;
        for (int k = 0; k < nm; k++) {
            size_t pos = rng() % t.size();
            switch (rng() % 4) {
            case 0: t[pos] = alphabet[rng() % (sizeof(alphabet) - 1)]; break;
            case 1: t.erase(pos, 1 + rng() % 5); break;
            case 2: t.insert(pos, 1, alphabet[rng() % (sizeof(alphabet) - 1)]); break;
            default: t.resize(pos); break;
            }
            if (t.empty()) t = "{";
        }
        std::string err;
        LinearCode lc; CssCode cc;
        bool r = (it & 1) ? linear_code_from_json(t.data(), t.size(), lc, err) : css_code_from_json(t.data(), t.size(), cc, err);
        if (r) ok++; else { bad++; if (err.empty()) { printf("empty error for: %s\n", t.c_str()); return 1; } }
    }
    printf("accepted %zu rejected %zu\n", ok, bad);
    return 0;
}
