// Issue cost of single VALU instructions on gfx950 (diagnostics, not product code): cycles per wave-instruction on one
// SIMD with 1 and 4 waves per SIMD, eight independent chains per lane.  Behind the instruction budgets of the flip
// kernel (Philox multiplies, bit searches, DPP reductions).
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_ops scripts/ubench_ops.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP> __global__ __launch_bounds__(256) void op_kernel(uint32_t iters, uint32_t *out, unsigned long long *cyc)
{
    uint32_t r[8];
    uint64_t w[8];
    for (int i = 0; i < 8; i++) {
        r[i] = threadIdx.x * 2654435761u + i * 40503u + 1;
        w[i] = ((uint64_t)r[i] << 32) | (r[i] ^ 0x5555u);
    }
    const uint32_t k = 0xD2511F53u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; it++) {
#define BODY(i)                                                                                                     \
    if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(k));                                       \
    if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(k));                                    \
    if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(k));                                    \
    if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(r[i]), "v"(k) : "vcc");       \
    if (OP == 4) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(k));                                   \
    if (OP == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(k));                                  \
    if (OP == 6) asm volatile("v_ffbl_b32 %0, %0" : "+v"(r[i]));                                                   \
    if (OP == 7) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(k));                                  \
    if (OP == 8) asm volatile("v_bfe_u32 %0, %0, %1, 7" : "+v"(r[i]) : "v"(k));                                    \
    if (OP == 9) asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i])); \
    if (OP == 10) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r[i]) : "v"(k));                               \
    if (OP == 11) asm volatile("v_log_f32 %0, %0" : "+v"(r[i]));                                                   \
    if (OP == 12) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[i]));                                               \
    if (OP == 13) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(k));                               \
    if (OP == 14) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(w[i]));                                            \
    if (OP == 15) asm volatile("v_mov_b32_dpp %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]));
        REP8(BODY)
        REP8(BODY)
#undef BODY
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++) acc ^= r[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kfn)(uint32_t, uint32_t *, unsigned long long *);

int main()
{
    uint32_t *out;
    unsigned long long *cyc;
    const int blocks = 256 * 4;
    if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess || hipMalloc(&cyc, blocks * 8) != hipSuccess) return 1;
    const char *names[16] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_u32_u24", "v_xor_b32", "v_ffbl_b32",
                             "v_bcnt_u32_b32", "v_bfe_u32", "s_nop 1 + v_min_u32_dpp", "v_lshl_or_b32", "v_log_f32", "v_cvt_f32_u32",
                             "v_mul_hi_u32_u24", "v_lshlrev_b64", "v_mov_b32_dpp"};
    kfn fns[16] = {op_kernel<0>, op_kernel<1>, op_kernel<2>, op_kernel<3>, op_kernel<4>, op_kernel<5>, op_kernel<6>, op_kernel<7>,
                   op_kernel<8>, op_kernel<9>, op_kernel<10>, op_kernel<11>, op_kernel<12>, op_kernel<13>, op_kernel<14>, op_kernel<15>};
    const uint32_t iters = 2000;
    std::vector<unsigned long long> h(blocks);
    for (int wps = 1; wps <= 4; wps *= 4) {
        // blocks of 256 threads = one wave per SIMD of a CU; `wps` blocks per CU
        const int nb = 256 * wps;
        for (int op = 0; op < 16; op++) {
            hipLaunchKernelGGL(fns[op], dim3(nb), dim3(256), 0, 0, iters, out, cyc);
            hipLaunchKernelGGL(fns[op], dim3(nb), dim3(256), 0, 0, iters, out, cyc);
            if (hipDeviceSynchronize() != hipSuccess) return 2;
            if (hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
            double s = 0;
            for (int i = 0; i < nb; i++) s += (double)h[i];
            // cycles per wave-instruction on its SIMD = block cycles / (instructions of one wave) / (waves per SIMD)
            printf("%d wave(s)/SIMD  %-26s %.2f cycles per wave-instruction per SIMD\n", wps, names[op],
                   s / nb / (16.0 * iters) / wps);
        }
    }
    return 0;
}
