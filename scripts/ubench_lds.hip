// LDS cost of the flip kernel's access kinds (diagnostics, not product code): 64 lanes at random dword addresses of a
// 150 KB region, 16 waves per CU, for ds_read_b32, ds_or_b32 (no return), ds_or_rtn_b32 and reads where 16 lanes share
// an address (the connecting-check reads of the set-up's pass 2).  Prints cycles per wave-instruction per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_lds scripts/ubench_lds.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

template <int OP> __global__ __launch_bounds__(1024) void lds_kernel(uint32_t iters, uint32_t *out, unsigned long long *cyc)
{
    extern __shared__ uint32_t s[];
    const uint32_t words = 150 * 256;
    for (uint32_t i = threadIdx.x; i < words; i += 1024) s[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = threadIdx.x * 2246822519u + 12345u, acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x = x * 1664525u + 1013904223u;
            uint32_t a = (x >> 8) % words;
            if (OP == 3) a = __shfl(a, threadIdx.x & 48, 64);   // 16 lanes share an address: 4 distinct addresses per wave
            if (OP == 0 || OP == 3) acc ^= s[a];
            if (OP == 1) atomicOr(&s[a], 1u << (x & 31));
            if (OP == 2) acc ^= atomicOr(&s[a], 1u << (x & 31));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    uint32_t *out;
    unsigned long long *cyc;
    const int blocks = 256;
    if (hipMalloc(&out, (size_t)blocks * 1024 * 4) != hipSuccess || hipMalloc(&cyc, blocks * 8) != hipSuccess) return 1;
    const char *names[4] = {"ds_read_b32, random", "ds_or_b32, random", "ds_or_rtn_b32, random", "ds_read_b32, 16 lanes per address"};
    typedef void (*kfn)(uint32_t, uint32_t *, unsigned long long *);
    kfn fns[4] = {lds_kernel<0>, lds_kernel<1>, lds_kernel<2>, lds_kernel<3>};
    const uint32_t iters = 500;
    std::vector<unsigned long long> h(blocks);
    for (int op = 0; op < 4; op++) {
        if (hipFuncSetAttribute((const void *)fns[op], hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return 4;
        hipLaunchKernelGGL(fns[op], dim3(blocks), dim3(1024), 150 * 1024, 0, iters, out, cyc);
        hipLaunchKernelGGL(fns[op], dim3(blocks), dim3(1024), 150 * 1024, 0, iters, out, cyc);
        if (hipDeviceSynchronize() != hipSuccess) return 2;
        if (hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
        double sum = 0;
        for (int i = 0; i < blocks; i++) sum += (double)h[i];
        // one block of 16 waves per CU; each wave issues 8 * iters LDS instructions
        printf("%-36s %.2f cycles per wave-instruction per CU (16 waves per CU)\n", names[op], sum / blocks / (8.0 * iters * 16));
    }
    return 0;
}
