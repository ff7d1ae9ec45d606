// examples/css_decoding.rs:7-45 against the C ABI (include/ldpc_hip.h).
// Build: hipcc -O2 -I../include css_decoding.cpp -L../ldpc_amd -lldpc_hip -Wl,-rpath,$PWD/../ldpc_amd -o css_decoding
#include <cstdio>
#include <cstdlib>

#include "ldpc_hip.h"

#define CHECK(call)                                                        \
    do {                                                                   \
        if ((call) != LDPC_OK) {                                           \
            std::fprintf(stderr, "%s: %s\n", #call, ldpc_last_error());    \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(int argc, char **argv)
{
    const size_t max_iterations = 100;                 // css_decoding.rs:8
    const double probability = 0.01;                   // css_decoding.rs:9
    const size_t shots = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1000;   // css_decoding.rs:14
    ldpc_code *base = nullptr;
    ldpc_css_code *code = nullptr;
    ldpc_css_bp_decoder *decoder = nullptr;
    CHECK(ldpc_code_random_regular(20, 15, 3, 4, /*seed*/ 1, &base));   // css_decoding.rs:26-32
    CHECK(ldpc_css_hypergraph_product(base, base, &code));              // css_decoding.rs:33
    CHECK(ldpc_css_bp_create(code, probability, max_iterations, &decoder));   // css_decoding.rs:36-45
    uint64_t counters[LDPC_COUNTERS];
    CHECK(ldpc_css_bp_mc_run(decoder, probability, /*seed*/ 2024, 0, shots, counters));   // css_decoding.rs:14-21
    std::printf("%llu\n", (unsigned long long)counters[LDPC_CNT_FAILURES]);   // css_decoding.rs:22
    ldpc_css_bp_destroy(decoder);
    ldpc_css_destroy(code);
    ldpc_code_destroy(base);
    return 0;
}
