// Micro-benchmarks behind the flip kernel's design numbers (diagnostics, not product code):
//   1. integer VALU issue rate per SIMD at 1 / 2 / 4 / 8 waves per SIMD
//   2. dependent LDS read latency (pointer chase), idle and under load
//   3. dependent L2-resident 16-byte load latency (pointer chase over a 2.5 MB table), idle and under load
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench scripts/ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ __launch_bounds__(256) void valu_int(uint32_t iters, uint32_t *out, unsigned long long *cyc)
{
    uint32_t a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = a * 5, f = a | 9, g = a + 11, h = a ^ 13;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < iters; i++) {
        a = (a << 1) ^ i; b = b + (i | 3); c = c ^ (c >> 3); d = (d & 0xFFFF) + i;
        e = e ^ (e << 5); f = f + 0x9E3779B9u; g = g ^ (g >> 7); h = (h | i) + 1;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(64) void lds_chase(uint32_t iters, uint32_t lds_words, uint32_t *out, unsigned long long *cyc)
{
    extern __shared__ uint32_t s[];
    for (uint32_t i = threadIdx.x; i < lds_words; i += 64) s[i] = (i * 97 + 13) % lds_words;
    __syncthreads();
    uint32_t p = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < iters; i++) p = s[p];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = p;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// every group of 16 lanes chases its own chain: one 256-byte row per hop (as the flip loop reads a neighbourhood row)
__global__ __launch_bounds__(64) void l2_chase(uint32_t iters, const uint4 *tab, uint32_t rows, uint32_t *out,
                                               unsigned long long *cyc)
{
    const uint32_t gl = threadIdx.x & 15;
    uint32_t r = (blockIdx.x * 4 + (threadIdx.x >> 4)) * 2654435761u % rows;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; i++) {
        const uint4 v = tab[(size_t)r * 16 + gl];
        acc ^= v.y;
        r = __shfl(v.x, threadIdx.x & 48, 64);   // next row = first word of the row
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = acc ^ r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static double avg(const std::vector<unsigned long long> &v, size_t n)
{
    double s = 0;
    for (size_t i = 0; i < n; i++) s += (double)v[i];
    return s / (double)n;
}

int main()
{
    uint32_t *out;
    unsigned long long *cyc;
    const size_t maxblocks = 256 * 32;
    CK(hipMalloc(&out, maxblocks * 256 * 4));
    CK(hipMalloc(&cyc, maxblocks * 8));
    std::vector<unsigned long long> h(maxblocks);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    printf("== integer VALU: 8 independent ops x iters, 256-thread blocks (4 waves = 1 per SIMD)\n");
    for (int bpc : {1, 2, 4, 8}) {
        const uint32_t iters = 20000, blocks = 256 * bpc;
        valu_int<<<blocks, 256>>>(100, out, cyc);
        CK(hipEventRecord(e0));
        valu_int<<<blocks, 256>>>(iters, out, cyc);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
        // the loop body compiles to ~14 VALU (8 ops, some with two instructions) -- report per source op
        printf("  %d waves/SIMD: %.3f ms, memtime ticks/wave %.0f, wall ns per (wave, source op) per SIMD %.3f\n", bpc, ms,
               avg(h, blocks), ms * 1e6 / ((double)iters * 8 * bpc));
    }

    printf("== LDS pointer chase (ds_read_b32 dependent chain), one-wave blocks\n");
    for (int bpc : {1, 4, 16}) {
        const uint32_t iters = 20000, blocks = 256 * bpc;
        CK(hipEventRecord(e0));
        lds_chase<<<blocks, 64, 8192>>>(iters, 2048, out, cyc);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
        printf("  %2d waves/CU: %.3f ms, ticks per hop %.1f, wall ns per hop %.1f\n", bpc, ms, avg(h, blocks) / iters,
               ms * 1e6 / iters);
    }

    printf("== L2 pointer chase: 256-B rows of a 2.56 MB table, 4 chains per wave\n");
    {
        const uint32_t rows = 10000;
        std::vector<uint4> tab((size_t)rows * 16);
        for (uint32_t r = 0; r < rows; r++)
            for (uint32_t k = 0; k < 16; k++) tab[(size_t)r * 16 + k] = uint4{(r * 7919u + 104729u) % rows, k, r, 0};
        uint4 *d;
        CK(hipMalloc(&d, tab.size() * 16));
        CK(hipMemcpy(d, tab.data(), tab.size() * 16, hipMemcpyHostToDevice));
        for (int bpc : {1, 4, 8, 16, 32}) {
            const uint32_t iters = 5000, blocks = 256 * bpc;
            l2_chase<<<blocks, 64>>>(100, d, rows, out, cyc);
            CK(hipEventRecord(e0));
            l2_chase<<<blocks, 64>>>(iters, d, rows, out, cyc);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
            printf("  %2d waves/CU: %.3f ms, ticks per hop %.1f, wall ns per hop %.1f, rows/s chip %.3e\n", bpc, ms,
                   avg(h, blocks) / iters, ms * 1e6 / iters, (double)blocks * 4 * iters / (ms * 1e-3));
        }
        CK(hipFree(d));
    }
    return 0;
}
