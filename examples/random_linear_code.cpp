// examples/random_linear_code.rs:5-17 against the C ABI.
#include <cstdio>

#include "ldpc_hip.h"

int main()
{
    ldpc_code *code = nullptr;
    if (ldpc_code_random_regular(4, 3, 3, 4, /*seed*/ 1, &code) != LDPC_OK) {
        std::fprintf(stderr, "%s\n", ldpc_last_error());
        return 1;
    }
    size_t positions[4], n = 0;
    if (ldpc_bsc_sample_error_of_length(0.2, ldpc_code_len(code), /*seed*/ 2024, /*shot*/ 0, positions, &n) != LDPC_OK) {
        std::fprintf(stderr, "%s\n", ldpc_last_error());
        return 1;
    }
    std::printf("[");
    for (size_t i = 0; i < n; i++) std::printf("%s%zu", i ? ", " : "", positions[i]);
    std::printf("]\n");
    ldpc_code_destroy(code);
    return 0;
}
