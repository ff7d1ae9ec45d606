// gfx950 kernels of the Monte-Carlo decoding path.
//
//   sample_*_kernel   noise (binary_symmetric_channel.rs:35-43, depolarizing.rs:34-48)
//   syndrome_kernel   H*e over GF(2) (linear_code/mod.rs:393-405)
//   flip_kernel       FlipDecoder::decode (flip.rs:21-43), fused with noise + syndrome + failure test
//   bp_kernel         BpDecoder::correction_for (belief_propagation.rs:23-30, 100-180), fused likewise
//
// Wavefront = 64 lanes everywhere; bit-packed vectors are built with 64-lane
// ballots, one u64 word per ballot.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "bp_math.hpp"
#include "kernels.hpp"

namespace ldpc {

namespace {

thread_local const char *g_kernel_name = "";   // host: the decoding kernel this thread launched last

constexpr int CNT_SHOTS = 0, CNT_FAIL = 1, CNT_DET = 2, CNT_UNDET = 3, CNT_WORK = 4, CNT_CAPPED = 5, CNT_XFAIL = 6,
              CNT_ZFAIL = 7;

// Lanes of one wave exchange data through LDS / global memory: order the
// accesses (the hardware executes one wave's LDS and vector-memory queues in
// order; this keeps the compiler from moving accesses across the hand-off and
// waits for outstanding stores).
__device__ inline void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ inline uint64_t wave_or64(uint64_t x)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)x, off, 64);
        uint32_t hi = __shfl_xor((uint32_t)(x >> 32), off, 64);
        x |= ((uint64_t)hi << 32) | lo;
    }
    return x;
}

__device__ inline uint64_t wave_sum64(uint64_t x)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)x, off, 64);
        uint32_t hi = __shfl_xor((uint32_t)(x >> 32), off, 64);
        x += ((uint64_t)hi << 32) | lo;
    }
    return x;
}

__device__ inline uint32_t wave_max_u32(uint32_t x)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = __shfl_xor(x, off, 64);
        x = o > x ? o : x;
    }
    return x;
}

__device__ inline uint64_t readlane64(uint64_t x, uint32_t l)
{
    uint32_t lo = __builtin_amdgcn_readlane((uint32_t)x, l);
    uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(x >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

__device__ inline uint32_t getbit(const uint64_t *w, uint32_t i) { return (uint32_t)(w[i >> 6] >> (i & 63)) & 1u; }

// ---------------------------------------------------------------------------
// Noise and syndrome, batched (shot-major packed words in HBM)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_bsc_kernel(GapTable gap, uint32_t n, uint64_t seed, uint64_t first,
                                                         uint64_t n_shots, uint64_t *words)
{
    __shared__ uint32_t q[65];
    for (uint32_t i = threadIdx.x; i < 65; i += blockDim.x) q[i] = gap.q[i];
    __syncthreads();
    const uint32_t Wn = (n + 63) / 64;
    const uint64_t total = n_shots * Wn;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t i = t / Wn;
        uint32_t w = (uint32_t)(t - i * Wn);
        words[t] = bsc_word(q, gap.inv_log2q, gap.p_is_zero, seed, first + i, w, bits_in_word(n, w));
    }
}

__global__ __launch_bounds__(256) void sample_depol_kernel(GapTable gap, uint32_t n, uint64_t seed, uint64_t first,
                                                           uint64_t n_shots, uint64_t *xw, uint64_t *zw)
{
    __shared__ uint32_t q[65];
    for (uint32_t i = threadIdx.x; i < 65; i += blockDim.x) q[i] = gap.q[i];
    __syncthreads();
    const uint32_t Wn = (n + 63) / 64;
    const uint64_t total = n_shots * Wn;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t i = t / Wn;
        uint32_t w = (uint32_t)(t - i * Wn);
        uint64_t x, z;
        depol_word(q, gap.inv_log2q, gap.p_is_zero, seed, first + i, w, bits_in_word(n, w), x, z);
        xw[t] = x;
        zw[t] = z;
    }
}

// H*e, sparse form: one wave per shot, the syndrome accumulates in the wave's LDS slice; the lanes walk the set
// bits of their words and toggle the checks of each (bit_adjacencies row) with LDS atomics.  Work follows the
// weight of e (C2 at p = 0.02: 200 bits x 3 checks) instead of the m x dc gathers of the row-wise form below.
__global__ __launch_bounds__(256) void syndrome_sparse_kernel(DeviceGraph g, uint64_t n_shots, const uint64_t *words,
                                                              uint64_t *synd)
{
    extern __shared__ uint64_t lds[];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t Wn = (g.n + 63) / 64, Wm = (g.m + 63) / 64;
    uint64_t *S = lds + (size_t)wv * Wm;
    uint32_t *S32 = (uint32_t *)S;
    for (uint64_t i = (uint64_t)blockIdx.x * nw + wv; i < n_shots; i += (uint64_t)gridDim.x * nw) {
        for (uint32_t w = lane; w < Wm; w += 64) S[w] = 0;
        wave_sync();
        for (uint32_t w = lane; w < Wn; w += 64) {
            uint64_t x = words[i * Wn + w];
            const uint32_t nb = bits_in_word(g.n, w);
            if (nb < 64) x &= (1ull << nb) - 1;
            while (x) {
                const uint32_t b = (w << 6) + (uint32_t)__builtin_ctzll(x);
                x &= x - 1;
                for (uint32_t k = g.t_ptr[b]; k < g.t_ptr[b + 1]; k++) {
                    const uint32_t c = g.t_idx[k];
                    atomicXor(&S32[c >> 5], 1u << (c & 31));
                }
            }
        }
        wave_sync();
        for (uint32_t w = lane; w < Wm; w += 64) synd[i * Wm + w] = S[w];
        wave_sync();
    }
}

// row-wise form (syndromes larger than LDS): one wave per shot, one lane per check row, 64 parities -> one ballot word
__global__ __launch_bounds__(256) void syndrome_kernel(DeviceGraph g, uint64_t n_shots, const uint64_t *words,
                                                       uint64_t *synd)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t Wn = (g.n + 63) / 64, Wm = (g.m + 63) / 64;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < n_shots; i += waves) {
        const uint64_t *e = words + i * Wn;
        for (uint32_t base = 0; base < g.m; base += 64) {
            uint32_t c = base + lane, par = 0;
            if (c < g.m)
                for (uint32_t k = g.h_ptr[c]; k < g.h_ptr[c + 1]; k++) par ^= getbit(e, g.h_idx[k]);
            uint64_t mask = __ballot(par);
            if (lane == 0) synd[i * Wm + (base >> 6)] = mask;
        }
    }
}

// test probe for bp_math.hpp: kind 0 = bp_tanh_half, 1 = bp_log1p, 2 / 3 = OCML's tanh(x/2) / log1p,
// 4 = bp_div_core(in[t], in[n + t]), 5 = bp_ref_atanh, 6 = bp_div_prod(in[t], in[n + t])
__global__ __launch_bounds__(256) void bp_math_probe_kernel(int kind, uint64_t n, const double *in, double *out)
{
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x) {
        double v;
        switch (kind) {
        case 0: v = bp_tanh_half(in[t]); break;
        case 1: v = bp_log1p(in[t]); break;
        case 2: v = tanh(in[t] / 2.0); break;
        case 3: v = log1p(in[t]); break;
        case 4: v = bp_div_core(in[t], in[n + t]); break;
        case 5: v = bp_ref_atanh(in[t]); break;
        case 7: case 8: case 9: case 10: {
            // bp_tanh_half_n, the batch form of bp_tanh_half, on the eight values in[8 (t / 8)..]: the result of element t % 8,
            // or NaN's stand-in -7.0 where the count leaves it out.  7 / 8: the scalar-constant form with a count that is
            // the same for the whole wave (blockIdx-dependent) / differs from lane to lane; 9 / 10: the plain form likewise.
            double x8[8], y8[8];
            const uint64_t base = t & ~(uint64_t)7;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                x8[j] = base + j < n ? in[base + j] : 0.0;
                y8[j] = -7.0;
            }
            const int cnt = (kind & 1) ? (int)((blockIdx.x + (t >> 8)) % 9) : (int)((t >> 3) % 9);
            if (kind <= 8) bp_tanh_half_n<8, true>(x8, y8, cnt);
            else bp_tanh_half_n<8, false>(x8, y8, cnt);
            v = y8[t & 7];
            out[n + t] = (double)cnt;   // (the second half of out: the count that was used)
            break;
        }
        default: v = bp_div_prod(in[t], in[n + t]); break;
        }
        out[t] = v;
    }
}

__global__ __launch_bounds__(256) void count_flags_kernel(const uint8_t *flags, const uint8_t *flags2, uint64_t n,
                                                          unsigned long long *counters)
{
    uint64_t f = 0, x = 0, z = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x) {
        uint8_t v = flags[t];
        if (flags2) v |= flags2[t];   // the two halves of a CSS run, when they ran side by side
        f += v != 0;
        x += v & 1;
        z += (v >> 1) & 1;
    }
    f = wave_sum64(f); x = wave_sum64(x); z = wave_sum64(z);
    if ((threadIdx.x & 63) == 0) {
        if (f) atomicAdd(&counters[CNT_FAIL], (unsigned long long)f);
        if (x) atomicAdd(&counters[CNT_XFAIL], (unsigned long long)x);
        if (z) atomicAdd(&counters[CNT_ZFAIL], (unsigned long long)z);
    }
}

// ---------------------------------------------------------------------------
// Flip decoder.  One wave per shot; the shot's state lives in LDS:
//   S[Wm]  syndrome, F[Wn] "flippable" mask (unsat > deg/2), O[Wn] output word.
// The reference rescans from bit 0 after every flip (flip.rs:27,35-43); the
// choice depends only on the current syndrome, so keeping F up to date for the
// bits whose checks changed and taking its lowest set bit is the same decision.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flip_kernel(DeviceGraph g, FlipArgs a)
{
    extern __shared__ uint64_t lds[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const uint32_t Wn = (g.n + 63) / 64, Wm = (g.m + 63) / 64;
    uint32_t *q = (uint32_t *)lds;   // 65 u32 in the first 33 u64
    for (uint32_t i = threadIdx.x; i < 65; i += blockDim.x) q[i] = a.gap.q[i];
    __syncthreads();
    uint64_t *S = lds + 33 + (size_t)wave * (Wm + 2 * Wn);
    uint64_t *F = S + Wm;
    uint64_t *O = F + Wn;
    uint32_t *S32 = (uint32_t *)S, *F32 = (uint32_t *)F;

    const uint64_t gw = (uint64_t)blockIdx.x * wpb + wave, waves = (uint64_t)gridDim.x * wpb;
    uint64_t c_shots = 0, c_fail = 0, c_det = 0, c_undet = 0, c_flips = 0;

    for (uint64_t i = gw; i < a.n_shots; i += waves) {
        const uint64_t shot = a.first_shot + i;
        // 1. error word (noise on device, or the caller's words)
        for (uint32_t w = lane; w < Wn; w += 64) {
            uint32_t nb = bits_in_word(g.n, w);
            uint64_t x;
            if (a.source == SRC_BSC) x = bsc_word(q, a.gap.inv_log2q, a.gap.p_is_zero, a.seed, shot, w, nb);
            else {
                x = a.words_in[i * Wn + w];
                if (nb < 64) x &= (1ull << nb) - 1;
            }
            O[w] = x;
            if (a.errors_out) a.errors_out[i * Wn + w] = x;
        }
        wave_sync();
        // 2. syndrome: one lane per check, ballot packs 64 checks
        for (uint32_t base = 0; base < g.m; base += 64) {
            uint32_t c = base + lane, par = 0;
            if (c < g.m)
                for (uint32_t k = g.h_ptr[c]; k < g.h_ptr[c + 1]; k++) par ^= getbit(O, g.h_idx[k]);
            uint64_t mask = __ballot(par);
            if (lane == 0) {
                S[base >> 6] = mask;
                if (a.synd_out) a.synd_out[i * Wm + (base >> 6)] = mask;
            }
        }
        wave_sync();
        // 3. flippable mask: one lane per bit
        for (uint32_t base = 0; base < g.n; base += 64) {
            uint32_t b = base + lane, flag = 0;
            if (b < g.n) {
                uint32_t k0 = g.t_ptr[b], k1 = g.t_ptr[b + 1], unsat = 0;
                for (uint32_t k = k0; k < k1; k++) unsat += getbit(S, g.t_idx[k]);
                flag = unsat > (k1 - k0) / 2;   // flip.rs:41
            }
            uint64_t mask = __ballot(flag);
            if (lane == 0) F[base >> 6] = mask;
        }
        wave_sync();
        // 4. serial flips, lowest index first
        uint32_t nflips = 0;
        for (;;) {
            int64_t bsel = -1;
            for (uint32_t wbase = 0; wbase < Wn; wbase += 64) {
                uint32_t w = wbase + lane;
                uint64_t x = w < Wn ? F[w] : 0;
                uint64_t nz = __ballot(x != 0);
                if (nz) {
                    uint32_t l = (uint32_t)__builtin_ctzll(nz);
                    uint64_t xw = readlane64(x, l);
                    bsel = (int64_t)(wbase + l) * 64 + __builtin_ctzll(xw);
                    break;
                }
            }
            if (bsel < 0) break;                                           // flip.rs:27 None
            const uint32_t b = (uint32_t)bsel;
            nflips++;
            const uint32_t k0 = g.t_ptr[b], deg = g.t_ptr[b + 1] - k0;
            if (lane == 0) O[b >> 6] ^= 1ull << (b & 63);                  // flip.rs:30
            for (uint32_t kk = lane; kk < deg; kk += 64) {                 // flip.rs:29 syndrome ^= H e_b
                uint32_t c = g.t_idx[k0 + kk];
                atomicXor(&S32[c >> 5], 1u << (c & 31));
            }
            wave_sync();
            // refresh F for every bit that shares a check with b
            for (uint32_t i2 = 0; i2 < deg; i2++) {
                uint32_t c = g.t_idx[k0 + i2];
                uint32_t e0 = g.h_ptr[c], dc = g.h_ptr[c + 1] - e0;
                for (uint32_t j = lane; j < dc; j += 64) {
                    uint32_t bb = g.h_idx[e0 + j];
                    uint32_t kk0 = g.t_ptr[bb], kk1 = g.t_ptr[bb + 1], unsat = 0;
                    for (uint32_t k = kk0; k < kk1; k++) unsat += getbit(S, g.t_idx[k]);
                    uint32_t bitm = 1u << (bb & 31);
                    if (unsat > (kk1 - kk0) / 2) atomicOr(&F32[bb >> 5], bitm);
                    else atomicAnd(&F32[bb >> 5], ~bitm);
                }
            }
            wave_sync();
        }
        // 5. failure test (all-zero codeword) and outputs
        uint64_t anyo = 0, anys = 0;
        for (uint32_t w = lane; w < Wn; w += 64) {
            uint64_t x = O[w];
            anyo |= x;
            if (a.words_out) a.words_out[i * Wn + w] = x;
        }
        for (uint32_t w = lane; w < Wm; w += 64) anys |= S[w];
        bool fail = __ballot(anyo != 0) != 0, det = __ballot(anys != 0) != 0;
        if (a.flips_out && lane == 0) a.flips_out[i] = nflips;
        c_shots++; c_fail += fail; c_det += det; c_undet += fail && !det; c_flips += nflips;
        wave_sync();
    }
    if (lane == 0 && a.counters) {
        atomicAdd(&a.counters[CNT_SHOTS], (unsigned long long)c_shots);
        if (c_fail) atomicAdd(&a.counters[CNT_FAIL], (unsigned long long)c_fail);
        if (c_det) atomicAdd(&a.counters[CNT_DET], (unsigned long long)c_det);
        if (c_undet) atomicAdd(&a.counters[CNT_UNDET], (unsigned long long)c_undet);
        if (c_flips) atomicAdd(&a.counters[CNT_WORK], (unsigned long long)c_flips);
    }
}

// ---------------------------------------------------------------------------
// Flip decoder for codes whose per-shot state does not fit the LDS of a CU (n = 10^6: 312 KB): the same wave-per-shot
// algorithm as flip_kernel with S, F and O in HBM / L2 (`big` scratch, one slice per wave) and two summary levels
// over F so that the lowest flippable bit costs three dependent reads, not a scan: FS bit w = F word w may be
// non-zero, FSS bit v = FS word v may be non-zero.  A summary bit can be stale (set over an empty word): the scan
// clears it and goes on.  Only this wave touches its slice; lanes meet through device-scope atomics and loads (a
// plain load could be served by a vector-L1 line that an atomic of another lane has since changed in L2).
// ---------------------------------------------------------------------------
__device__ inline uint64_t ld64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint32_t getbit_g(const uint64_t *w, uint32_t i) { return (uint32_t)(ld64(w + (i >> 6)) >> (i & 63)) & 1u; }
__device__ inline void xorbit_g(uint64_t *w, uint32_t i) { atomicXor((unsigned long long *)(w + (i >> 6)), 1ull << (i & 63)); }
__device__ inline void orbit_g(uint64_t *w, uint32_t i) { atomicOr((unsigned long long *)(w + (i >> 6)), 1ull << (i & 63)); }
__device__ inline void clrbit_g(uint64_t *w, uint32_t i) { atomicAnd((unsigned long long *)(w + (i >> 6)), ~(1ull << (i & 63))); }

__host__ __device__ inline size_t flip_big_words(uint32_t n, uint32_t m)
{
    const size_t Wn = (n + 63) / 64, Wm = (m + 63) / 64, W1 = (Wn + 63) / 64, W2 = (W1 + 63) / 64;
    return Wm + 2 * Wn + W1 + W2 + 8;
}

#ifndef FLIP_BIG_WIDE
#define FLIP_BIG_WIDE 1
#endif
__global__ __launch_bounds__(256, 8) void flip_big_kernel(DeviceGraph g, FlipArgs a, uint64_t *big)
{
    __shared__ uint32_t q[65];
    for (uint32_t i = threadIdx.x; i < 65; i += blockDim.x) q[i] = a.gap.q[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const uint32_t Wn = (g.n + 63) / 64, Wm = (g.m + 63) / 64, W1 = (Wn + 63) / 64, W2 = (W1 + 63) / 64;
    const uint64_t gw = (uint64_t)blockIdx.x * wpb + wave, waves = (uint64_t)gridDim.x * wpb;
    uint64_t *S = big + gw * flip_big_words(g.n, g.m), *F = S + Wm, *O = F + Wn, *FS = O + Wn, *FSS = FS + W1;
    const uint32_t state_words = Wm + 2 * Wn + W1 + W2;
    uint64_t c_shots = 0, c_fail = 0, c_det = 0, c_undet = 0, c_flips = 0;
    const bool small = g.max_dv <= 8 && g.max_dc <= 8 && FLIP_BIG_WIDE;   // (wave-uniform: the flip step's parallel form)

    // unsatisfied checks of bit b > deg / 2 (flip.rs:41)
    auto flippable = [&](uint32_t b) -> bool {
        const uint32_t k0 = g.t_ptr[b], k1 = g.t_ptr[b + 1];
        uint32_t unsat = 0;
        for (uint32_t k = k0; k < k1; k++) unsat += getbit_g(S, g.t_idx[k]);
        return unsat > (k1 - k0) / 2;
    };
    auto set_f = [&](uint32_t b) {
        orbit_g(F, b);
        orbit_g(FS, b >> 6);
        orbit_g(FSS, b >> 12);
    };

    for (uint64_t i = gw; i < a.n_shots; i += waves) {
        const uint64_t shot = a.first_shot + i;
        for (uint32_t w = lane; w < state_words; w += 64) S[w] = 0;
        // 1. error word -> O, syndrome by toggling the checks of every error bit
        for (uint32_t w = lane; w < Wn; w += 64) {
            const uint32_t nb = bits_in_word(g.n, w);
            uint64_t x;
            if (a.source == SRC_BSC) x = bsc_word(q, a.gap.inv_log2q, a.gap.p_is_zero, a.seed, shot, w, nb);
            else {
                x = a.words_in[i * Wn + w];
                if (nb < 64) x &= (1ull << nb) - 1;
            }
            O[w] = x;
            if (a.errors_out) a.errors_out[i * Wn + w] = x;
        }
        wave_sync();   // the zeroes and O are in memory before the atomics below
        for (uint32_t w = lane; w < Wn; w += 64) {
            uint64_t x = ld64(O + w);
            while (x) {
                const uint32_t b = (w << 6) + (uint32_t)__builtin_ctzll(x);
                x &= x - 1;
                for (uint32_t k = g.t_ptr[b]; k < g.t_ptr[b + 1]; k++) xorbit_g(S, g.t_idx[k]);
            }
        }
        wave_sync();
        if (a.synd_out)
            for (uint32_t w = lane; w < Wm; w += 64) a.synd_out[i * Wm + w] = ld64(S + w);
        // 2. flippable mask: only a bit on an unsatisfied check can be flippable
        for (uint32_t w = lane; w < Wm; w += 64) {
            uint64_t x = ld64(S + w);
            while (x) {
                const uint32_t c = (w << 6) + (uint32_t)__builtin_ctzll(x);
                x &= x - 1;
                for (uint32_t e = g.h_ptr[c]; e < g.h_ptr[c + 1]; e++) {
                    const uint32_t b = g.h_idx[e];
                    if (flippable(b)) set_f(b);
                }
            }
        }
        wave_sync();
        // 3. serial flips, lowest index first (flip.rs:27,35-43); at most m flips, each lowers the syndrome weight
        uint32_t nflips = 0;
        for (uint64_t guard = 0; guard < 40ull * g.m + 4ull * Wn + 64; guard++) {   // (flips + stale summary bits; never spin on a bug)
            // first set bit of FSS -> FS word -> F word
            int32_t v = -1;
            for (uint32_t base = 0; base < W2 && v < 0; base += 64) {
                const uint64_t x = base + lane < W2 ? ld64(FSS + base + lane) : 0;
                const uint64_t nz = __ballot(x != 0);
                if (nz) {
                    const uint32_t l = (uint32_t)__builtin_ctzll(nz);
                    v = (int32_t)((base + l) * 64 + (uint32_t)__builtin_ctzll(readlane64(x, l)));
                }
            }
            if (v < 0) break;                                              // find_flippable == None
            const uint64_t fs = ld64(FS + v);
            if (!fs) {                                                     // stale: nothing under this FS word
                if (lane == 0) clrbit_g(FSS, (uint32_t)v);
                wave_sync();
                continue;
            }
            const uint32_t w = (uint32_t)v * 64 + (uint32_t)__builtin_ctzll(fs);
            const uint64_t fw = ld64(F + w);
            if (!fw) {                                                     // stale: the F word emptied since
                if (lane == 0) clrbit_g(FS, w);
                wave_sync();
                continue;
            }
            const uint32_t b = w * 64 + (uint32_t)__builtin_ctzll(fw);
            nflips++;
            const uint32_t k0 = g.t_ptr[b], deg = g.t_ptr[b + 1] - k0;
            if (lane == 0) xorbit_g(O, b);                                 // flip.rs:30
            if (small) {
                // Degrees up to eight (round 3): the neighbourhood of b is refreshed in one go, lane = (check of b, bit of
                // that check), and a neighbour's checks and syndrome words are asked for together -- a dozen dependent
                // memory round trips per flip where the general form below takes about forty (three checks one after
                // the other, each neighbour's checks one after the other).  A neighbour met through two checks is
                // refreshed twice, to the same value.
                uint32_t e0_l = 0, dc_l = 0;
                if (lane < deg) {
                    const uint32_t c = g.t_idx[k0 + lane];
                    xorbit_g(S, c);                                        // flip.rs:29
                    e0_l = g.h_ptr[c];
                    dc_l = g.h_ptr[c + 1] - e0_l;
                }
                wave_sync();
                const uint32_t i2 = lane >> 3, j = lane & 7;
                const uint32_t e0 = (uint32_t)__shfl((int)e0_l, (int)i2, 64), dc = (uint32_t)__shfl((int)dc_l, (int)i2, 64);
                if (i2 < deg && j < dc) {
                    const uint32_t bb = g.h_idx[e0 + j];
                    const uint32_t kb0 = g.t_ptr[bb], degb = g.t_ptr[bb + 1] - kb0;
                    uint32_t ck[8];
                    uint64_t sw[8];
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) ck[k] = k < degb ? g.t_idx[kb0 + k] : 0u;
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) sw[k] = k < degb ? ld64(S + (ck[k] >> 6)) : 0ull;
                    uint32_t unsat = 0;
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) unsat += (uint32_t)(sw[k] >> (ck[k] & 63)) & (k < degb ? 1u : 0u);
                    if (unsat > degb / 2) set_f(bb);
                    else clrbit_g(F, bb);
                }
                wave_sync();
                continue;
            }
            for (uint32_t kk = lane; kk < deg; kk += 64) xorbit_g(S, g.t_idx[k0 + kk]);   // flip.rs:29
            wave_sync();
            // refresh F for every bit that shares a check with b (b itself included)
            for (uint32_t i2 = 0; i2 < deg; i2++) {
                const uint32_t c = g.t_idx[k0 + i2];
                const uint32_t e0 = g.h_ptr[c], dc = g.h_ptr[c + 1] - e0;
                for (uint32_t j = lane; j < dc; j += 64) {
                    const uint32_t bb = g.h_idx[e0 + j];
                    if (flippable(bb)) set_f(bb);
                    else clrbit_g(F, bb);
                }
            }
            wave_sync();
        }
        // 4. failure test (all-zero codeword) and outputs
        uint64_t anyo = 0, anys = 0;
        for (uint32_t w = lane; w < Wn; w += 64) {
            const uint64_t x = ld64(O + w);
            anyo |= x;
            if (a.words_out) a.words_out[i * Wn + w] = x;
        }
        for (uint32_t w = lane; w < Wm; w += 64) anys |= ld64(S + w);
        const bool fail = __ballot(anyo != 0) != 0, det = __ballot(anys != 0) != 0;
        if (a.flips_out && lane == 0) a.flips_out[i] = nflips;
        c_shots++; c_fail += fail; c_det += det; c_undet += fail && !det; c_flips += nflips;
        wave_sync();
    }
    if (lane == 0 && a.counters) {
        atomicAdd(&a.counters[CNT_SHOTS], (unsigned long long)c_shots);
        if (c_fail) atomicAdd(&a.counters[CNT_FAIL], (unsigned long long)c_fail);
        if (c_det) atomicAdd(&a.counters[CNT_DET], (unsigned long long)c_det);
        if (c_undet) atomicAdd(&a.counters[CNT_UNDET], (unsigned long long)c_undet);
        if (c_flips) atomicAdd(&a.counters[CNT_WORK], (unsigned long long)c_flips);
    }
}

// ---------------------------------------------------------------------------
// Flip decoder, fast path.  One wave carries FOUR shots: the serial flip loop
// of a shot only ever touches the <= K bits around the flipped bit, so a group
// of 16 lanes per shot runs it (K slots = one or more rounds of 16), and the
// four groups of a wave advance four independent serial chains in lockstep.
// The set-up of a shot (noise, syndrome, initial flippable mask) is done
// wave-wide and sparsely: the sampled error positions are compacted into an LDS
// list, pass 1 toggles the checks of every error bit, pass 2 evaluates only the
// bits in the neighbourhood of an error bit (all other bits see no unsatisfied
// check and cannot be flippable).
// ---------------------------------------------------------------------------
constexpr uint32_t NONE = 0xFFFFFFFFu;
// LDS header of a flip2 block in u64 words: gap table (33), the error-position list of the shot being set up
// (list_cap u32 entries; a power of two, 64..1024, chosen by the launcher from the expected error weight: LDS
// bounds the blocks per CU, and a shot with more errors than the list holds takes the piecewise path), counter
__host__ __device__ constexpr uint32_t flip2_hdr_words(uint32_t list_cap) { return 33 + list_cap / 2 + 1; }

__device__ inline uint32_t row_min_u32(uint32_t v);
__device__ inline uint32_t half_row_min_u32(uint32_t v);
template <int G> __device__ inline uint32_t group_min_u32(uint32_t v)
{
    if (G == 16) return row_min_u32(v);
    return half_row_min_u32(v);
}

template <int DVP> struct alignas(16) NbRec {
    uint32_t bit;
    uint32_t chk[DVP];
};

// chk fields of a record are packed: (byte offset of the check's 32-bit syndrome word) << 16 | bit in that word
__device__ inline uint32_t *chk_word(uint32_t *S32, uint32_t chk) { return (uint32_t *)((char *)S32 + (chk >> 16)); }
__device__ inline uint32_t chk_mask(uint32_t chk) { return 1u << (chk & 31); }

template <int DVP, bool REG> __device__ inline bool rec_flippable(const NbRec<DVP> &r, uint32_t *S32)
{
    // all S reads are issued back to back (one LDS wait), then reduced.  REG: every bit has
    // exactly DVP checks, no sentinel tests.
    uint32_t word[DVP];
#pragma unroll
    for (int k = 0; k < DVP; k++) word[k] = *chk_word(S32, (REG || r.chk[k] != NONE) ? r.chk[k] : 0u);
    uint32_t deg = 0, unsat = 0;
#pragma unroll
    for (int k = 0; k < DVP; k++) {
        const uint32_t valid = (REG || r.chk[k] != NONE) ? 1u : 0u;
        deg += valid;
        unsat += valid & (word[k] >> (r.chk[k] & 31));
    }
    return unsat > deg / 2;   // flip.rs:41
}

// Orders one wave's LDS traffic (wavefront-scope fence: no cache action, no wait for
// retirement): the LDS pipeline executes a wave's instructions in issue order, so a
// later read sees an earlier atomic of another lane of the same wave.
__device__ inline void wave_order()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// minimum over each group of 8 lanes: xor-1 and xor-2 quad permutes, then the half-row mirror
__device__ inline uint32_t half_row_min_u32(uint32_t v)
{
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);    // quad_perm:[1,0,3,2]
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);    // quad_perm:[2,3,0,1]
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    v = o < v ? o : v;
    return v;
}

// minimum over each row of 16 lanes (DPP row rotations: no LDS round trip)
__device__ inline uint32_t row_min_u32(uint32_t v)
{
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);   // row_ror:8
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x124, 0xF, 0xF, true);   // row_ror:4
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x122, 0xF, 0xF, true);   // row_ror:2
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x121, 0xF, 0xF, true);   // row_ror:1
    v = o < v ? o : v;
    return v;
}

// ONE: K == G, every row is exactly one round of the group (kshift becomes a constant, the round loop unrolls away)
// SMALL: one window of G summary words covers the whole flippable mask (n <= 1024 G): no scan cursor
template <int DVP, int G, bool REG, bool DIAG, bool ONE, bool SMALL> __global__ __launch_bounds__(64) void flip2_kernel(DeviceGraph g, NbTable nbt, FlipArgs a)
{
    extern __shared__ uint64_t lds[];
    constexpr uint32_t SPW = 64 / G;                 // shots per wave
    const uint32_t lane = threadIdx.x, grp = lane / G, gl = lane % G;
    const uint32_t Wn = (g.n + 63) / 64, Wm = (g.m + 63) / 64;
    const uint32_t WS = (2 * Wn + 63) / 64;          // u64 words of the summary FS (one bit per u32 word of F)
    const uint32_t WS32 = (2 * Wn + 31) / 32;        // FS as u32 words: bit w set if the u32 word w of F may be non-zero
    const uint32_t per_shot = Wm + Wn + WS;          // LDS words per shot: S, F, FS (O lives in HBM)
    uint32_t *q = (uint32_t *)lds;
    uint32_t *list = (uint32_t *)(lds + 33);         // error positions of the shot being set up
    const uint32_t LIST_CAP = a.list_cap;
    uint32_t *cnt_lds = (uint32_t *)(lds + 33 + LIST_CAP / 2);
    uint64_t *state = lds + flip2_hdr_words(LIST_CAP);
    const NbRec<DVP> *nb = (const NbRec<DVP> *)nbt.rec;
    const uint32_t kshift = ONE ? (G == 16 ? 4u : 3u) : nbt.kshift, K = 1u << kshift;

    for (uint32_t i = lane; i < 65; i += 64) q[i] = a.gap.q[i];
    wave_sync();

    uint64_t c_shots = 0, c_fail = 0, c_det = 0, c_undet = 0, c_flips = 0, c_iters = 0;

    for (uint64_t base = (uint64_t)blockIdx.x * SPW; base < a.n_shots; base += (uint64_t)gridDim.x * SPW) {
        const uint32_t nvalid = a.n_shots - base < SPW ? (uint32_t)(a.n_shots - base) : SPW;
        // the sampled / given word of shot j is parked in HBM during the flip loop (the caller's record
        // buffer, or this block's scratch slice); the flips are logged and applied to it at the end
        auto err_words = [&](uint32_t j) -> uint64_t * {
            return a.errors_out ? a.errors_out + (base + j) * Wn : a.o_scratch + ((size_t)blockIdx.x * SPW + j) * Wn;
        };
        // ------------------------------------------------------------------ set-up, wave-wide, shot by shot
        for (uint32_t j = 0; j < nvalid; j++) {
            uint64_t *S = state + (size_t)j * per_shot, *F = S + Wm, *O = err_words(j);
            uint32_t *S32 = (uint32_t *)S, *F32 = (uint32_t *)F, *FS32 = (uint32_t *)(F + Wn);
            const uint64_t i = base + j, shot = a.first_shot + i;
            for (uint32_t w = lane; w < Wm; w += 64) S[w] = 0;
            for (uint32_t w = lane; w < Wn + WS; w += 64) F[w] = 0;   // F and its summary FS
            if (lane == 0) *cnt_lds = 0;
            wave_order();
            // noise (or the caller's word); error positions go to the LDS list as they appear
            for (uint32_t w = lane; w < Wn; w += 64) {
                const uint32_t nb_bits = bits_in_word(g.n, w);
                uint64_t x;
                if (a.source == SRC_BSC) x = bsc_word(q, a.gap.inv_log2q, a.gap.p_is_zero, a.seed, shot, w, nb_bits);
                else {
                    x = a.words_in[i * Wn + w];
                    if (nb_bits < 64) x &= (1ull << nb_bits) - 1;
                }
                O[w] = x;
                const uint32_t cc = (uint32_t)__builtin_popcountll(x);
                if (cc) {
                    uint32_t off = atomicAdd(cnt_lds, cc);
                    while (x) {
                        if (off < LIST_CAP) list[off] = (w << 6) + (uint32_t)__builtin_ctzll(x);
                        off++;
                        x &= x - 1;
                    }
                }
            }
            wave_sync();   // O is read back from memory below (overflow path) and at the end
            const uint32_t total = __builtin_amdgcn_readfirstlane(*cnt_lds);
            if (total <= LIST_CAP) {
                // pass 1, syndrome: toggle the checks of every error bit (slot 0 of its row)
                for (uint32_t t0 = lane; t0 < total; t0 += 64 * 4) {
                    NbRec<DVP> r[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; u++) {
#pragma unroll
                        for (int k = 0; k < DVP; k++) r[u].chk[k] = NONE;
                        if (t0 + 64 * u < total) r[u] = nb[(size_t)list[t0 + 64 * u] << kshift];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 4; u++)
#pragma unroll
                        for (int k = 0; k < DVP; k++)
                            if (r[u].chk[k] != NONE) atomicXor(chk_word(S32, r[u].chk[k]), chk_mask(r[u].chk[k]));
                }
                wave_order();
                // pass 2, flippable mask: only bits next to an error bit can see an unsatisfied check
                // (four records per trip: their row loads are in flight together)
                constexpr uint32_t P2 = 8;
                for (uint32_t t0 = lane; t0 < (total << kshift); t0 += 64 * P2) {
                    NbRec<DVP> r[P2];
#pragma unroll
                    for (uint32_t u = 0; u < P2; u++) {
                        const uint32_t t = t0 + 64 * u;
                        r[u].bit = NONE;
                        if (t < (total << kshift)) r[u] = nb[((size_t)list[t >> kshift] << kshift) + (t & (K - 1))];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < P2; u++)
                        if (r[u].bit != NONE && rec_flippable<DVP, REG>(r[u], S32)) {
                            atomicOr(&F32[r[u].bit >> 5], 1u << (r[u].bit & 31));
                            atomicOr(&FS32[r[u].bit >> 10], 1u << ((r[u].bit >> 5) & 31));
                        }
                }
            } else {
                // more errors than the list holds: the same two passes over pieces of LIST_CAP / 64 words,
                // each piece (<= LIST_CAP positions) listed in turn
                const uint32_t pw = LIST_CAP / 64;
                for (int pass = 0; pass < 2; pass++) {
                    for (uint32_t wb = 0; wb < Wn; wb += pw) {
                        wave_order();
                        if (lane == 0) *cnt_lds = 0;
                        wave_order();
                        const uint32_t w = wb + lane;
                        uint64_t x = (lane < pw && w < Wn) ? O[w] : 0;
                        const uint32_t cc = (uint32_t)__builtin_popcountll(x);
                        if (cc) {
                            uint32_t off = atomicAdd(cnt_lds, cc);
                            while (x) {
                                list[off++] = (w << 6) + (uint32_t)__builtin_ctzll(x);
                                x &= x - 1;
                            }
                        }
                        wave_order();
                        const uint32_t cnt = __builtin_amdgcn_readfirstlane(*cnt_lds);
                        if (pass == 0) {
                            for (uint32_t t = lane; t < cnt; t += 64) {
                                const NbRec<DVP> r = nb[(size_t)list[t] << kshift];
#pragma unroll
                                for (int k = 0; k < DVP; k++)
                                    if (r.chk[k] != NONE) atomicXor(chk_word(S32, r.chk[k]), chk_mask(r.chk[k]));
                            }
                        } else {
                            for (uint32_t t = lane; t < (cnt << kshift); t += 64) {
                                const NbRec<DVP> r = nb[((size_t)list[t >> kshift] << kshift) + (t & (K - 1))];
                                if (r.bit != NONE && rec_flippable<DVP, REG>(r, S32)) {
                                    atomicOr(&F32[r.bit >> 5], 1u << (r.bit & 31));
                                    atomicOr(&FS32[r.bit >> 10], 1u << ((r.bit >> 5) & 31));
                                }
                            }
                        }
                    }
                    wave_order();
                }
            }
            wave_order();
            if (a.synd_out)
                for (uint32_t w = lane; w < Wm; w += 64) a.synd_out[i * Wm + w] = S[w];
        }
        // ------------------------------------------------------------------ serial flips, SPW shots in lockstep
        {
            uint64_t *S = state + (size_t)grp * per_shot, *F = S + Wm, *FS = F + Wn;
            uint32_t *S32 = (uint32_t *)S, *F32 = (uint32_t *)F, *FS32 = (uint32_t *)FS;
            // flip log of this group's shot, one entry per flip; at most m flips per shot (each lowers the
            // syndrome weight)
            uint32_t *flog = a.flip_log + ((size_t)blockIdx.x * SPW + grp) * a.flip_log_cap;
            bool active = grp < nvalid;
            uint32_t curS = 0, nfl = 0;   // invariant: no flippable bit in F words below 32*curS
            const uint32_t max_flips = a.max_flips;
            while (__ballot(active)) {
                // lowest flippable bit of the shot, two levels: the summary FS names the first F word that may
                // be non-zero (each lane looks at one summary word, the group takes the minimum), then that
                // F word gives the bit.  A summary bit can be stale (set, word empty): it is cleared and the
                // iteration does nothing else.
                const uint32_t sw = SMALL ? gl : curS + gl;
                const uint32_t fs = (active && sw < WS32) ? FS32[sw] : 0;
                uint32_t candw = fs ? (sw << 5) + (uint32_t)__builtin_ctz(fs) : NONE;
                candw = group_min_u32<G>(candw);
                const bool havew = active && candw != NONE;
                if (active && !havew) {                                 // flip.rs:27 find_flippable == None
                    if (SMALL) active = false;
                    else {
                        curS += G;
                        if (curS >= WS32) active = false;
                    }
                }
                const uint32_t x = havew ? F32[candw] : 0;
                const bool found = havew && x != 0;
                if (havew && !found && gl == 0) atomicAnd(&FS32[candw >> 5], ~(1u << (candw & 31)));
                const uint32_t b = found ? (candw << 5) + (uint32_t)__builtin_ctz(x) : 0u;
                uint32_t minset = NONE;
                // the row of b (row 0 for a group with nothing to flip: a harmless read, no divergence)
                NbRec<DVP> rec_next = nb[((size_t)b << kshift) + gl];
                for (uint32_t r = 0; r < K; r += G) {
                    const NbRec<DVP> rec = rec_next;
                    if (r + G < K) rec_next = nb[((size_t)b << kshift) + r + G + gl];   // next round's slots in flight
                    if (r == 0) {
                        // slot 0 (lane 0 of the group) is b itself: syndrome ^= H e_b (flip.rs:28-29)
                        if (found && gl == 0) {   // one lane per shot: the atomics of idle lanes would only collide
#pragma unroll
                            for (int k = 0; k < DVP; k++)
                                if (REG || rec.chk[k] != NONE) atomicXor(chk_word(S32, rec.chk[k]), chk_mask(rec.chk[k]));
                            // b was the only flippable bit of its word: the word empties (a neighbour that
                            // becomes flippable in it sets the summary bit again below)
                            if ((x & (x - 1)) == 0) atomicAnd(&FS32[candw >> 5], ~(1u << (candw & 31)));
                        }
                        wave_order();
                    }
                    // every slot re-evaluates its bit against the new syndrome and fixes its F bit
                    const bool have = found && rec.bit != NONE;
                    const uint32_t fword = F32[have ? (rec.bit >> 5) : 0];
                    const bool fl = have && rec_flippable<DVP, REG>(rec, S32);
                    const bool old = (fword >> (rec.bit & 31)) & 1u;
                    if (have && fl != old) {
                        atomicXor(&F32[rec.bit >> 5], 1u << (rec.bit & 31));
                        if (fl) atomicOr(&FS32[rec.bit >> 10], 1u << ((rec.bit >> 5) & 31));
                    }
                    if (!SMALL && fl) minset = rec.bit < minset ? rec.bit : minset;
                }
                // output ^= e_b (flip.rs:30) is deferred: log the flip
                if (found) {
                    if (nfl >= a.flip_log_cap) active = false;   // cannot happen (<= m flips); never write past the log
                    else if (gl == 0) flog[nfl] = b;
                    nfl++;
                }
                if (!SMALL) {   // more summary words than lanes: keep the scan cursor at the lowest possible word
                    minset = group_min_u32<G>(minset);
                    if (found) {
                        const uint32_t bs = b >> 10, ms = minset == NONE ? NONE : minset >> 10;
                        curS = ms < bs ? ms : bs;
                    }
                }
                wave_order();
                if (DIAG && active) c_iters++;                  // profiling aid: loop iterations incl. empty ones
                if (DIAG && nfl >= max_flips) active = false;   // profiling aid only
            }
            wave_sync();   // log entries are in memory
            // ---------------------------------------------------------------- failure test + outputs
            for (uint32_t j = 0; j < nvalid; j++) {
                uint64_t *Sj = state + (size_t)j * per_shot, *Oj = Sj + Wm;   // F is dead: rebuild the word there
                uint32_t *O32 = (uint32_t *)Oj;
                const uint64_t *Ej = err_words(j);
                const uint32_t *lj = a.flip_log + ((size_t)blockIdx.x * SPW + j) * a.flip_log_cap;
                const uint64_t i = base + j;
                const uint32_t nfj = __shfl(nfl, j * G, 64), nf = nfj < a.flip_log_cap ? nfj : a.flip_log_cap;
                // device-scope loads: both were written by this wave, possibly through another lane
                for (uint32_t w = lane; w < Wn; w += 64) Oj[w] = __hip_atomic_load(&Ej[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wave_order();
                for (uint32_t t0 = lane; t0 < nf; t0 += 64 * 4) {
                    uint32_t bb[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; u++)
                        bb[u] = t0 + 64 * u < nf ? __hip_atomic_load(&lj[t0 + 64 * u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : NONE;
#pragma unroll
                    for (uint32_t u = 0; u < 4; u++)
                        if (bb[u] != NONE) atomicXor(&O32[bb[u] >> 5], 1u << (bb[u] & 31));                // flip.rs:30
                }
                wave_order();
                uint64_t anyo = 0, anys = 0;
                for (uint32_t w = lane; w < Wn; w += 64) {
                    const uint64_t x = Oj[w];
                    anyo |= x;
                    if (a.words_out) a.words_out[i * Wn + w] = x;
                }
                for (uint32_t w = lane; w < Wm; w += 64) anys |= Sj[w];
                const bool fail = __ballot(anyo != 0) != 0, det = __ballot(anys != 0) != 0;
                if (a.flips_out && lane == 0) a.flips_out[i] = nf;
                c_shots++; c_fail += fail; c_det += det; c_undet += fail && !det; c_flips += nf;
            }
            wave_sync();
        }
    }
    if (lane == 0 && a.counters) {
        atomicAdd(&a.counters[CNT_SHOTS], (unsigned long long)c_shots);
        if (c_fail) atomicAdd(&a.counters[CNT_FAIL], (unsigned long long)c_fail);
        if (c_det) atomicAdd(&a.counters[CNT_DET], (unsigned long long)c_det);
        if (c_undet) atomicAdd(&a.counters[CNT_UNDET], (unsigned long long)c_undet);
        if (c_flips) atomicAdd(&a.counters[CNT_WORK], (unsigned long long)c_flips);
    }
    if (DIAG && a.counters && gl == 0 && c_iters) atomicAdd(&a.counters[CNT_CAPPED], (unsigned long long)c_iters);
}

// ---------------------------------------------------------------------------
// Flip decoder, third form: regular codes of bit degree 3 whose neighbourhoods
// fit 16 slots and whose flippable mask fits 32 windows of 512 bits (C1, C2).
// One wave carries EIGHT shots, 8 lanes each, every lane working two slots of
// the flipped bit's neighbourhood row, and the loop body is branch-free:
//   * the summary of the flippable mask F is one 32-bit word per shot kept in
//     registers (bit w = window w of F may hold a flippable bit), so finding the
//     lowest flippable bit is one 64-byte LDS read per shot and a DPP minimum;
//   * a slot's F bit is written with ds_mskor (clear-then-set in one LDS
//     operation, no read, no compare, no divergent or/and);
//   * padding slots and the rounds of a shot that has nothing left to flip
//     point at a spare check that is never set and a spare mask bit that is
//     never scanned, so they need no predication.
// Same decisions as flip.rs:21-43: F is exact after every flip and its lowest
// set bit is flipped next.
// ---------------------------------------------------------------------------
__device__ inline uint32_t lds_addr_of(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

// LDS: mem = (mem & ~mask) | data in one operation
__device__ inline void lds_mskor(uint32_t addr, uint32_t mask, uint32_t data)
{
    asm volatile("ds_mskor_b32 %0, %1, %2" ::"v"(addr), "v"(mask), "v"(data) : "memory");
}

// minimum over each group of 8 lanes, DPP folded into the v_min (every source lane is in range: bound_ctrl is moot)
__device__ inline uint32_t half_row_min2_u32(uint32_t v)
{
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);    // quad_perm:[1,0,3,2]
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);    // quad_perm:[2,3,0,1]
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v = o < v ? o : v;
    return v;
}

__device__ inline uint32_t half_row_add_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);
    return v;
}

__device__ inline uint32_t half_row_or_u32(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);    // quad_perm:[1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);    // quad_perm:[2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    return v;
}

// bit (pos & 31) of w in one instruction (v_bfe_u32 takes the low five bits of its offset operand by itself; written
// in C the compiler spends a second instruction on the `& 31` or on the `& 1`)
__device__ inline uint32_t bit_of(uint32_t w, uint32_t pos)
{
    return __builtin_amdgcn_ubfe(w, pos, 1u);
}

// unsatisfied checks of a record >= 2 (bit degree 3: flip.rs:41 `> deg / 2`)
__device__ inline bool rec3_flippable(const uint4 &r, const char *S)
{
    const uint32_t w0 = *(const uint32_t *)(S + (r.y >> 16)), w1 = *(const uint32_t *)(S + (r.z >> 16)),
                   w2 = *(const uint32_t *)(S + (r.w >> 16));
    return bit_of(w0, r.y) + bit_of(w1, r.z) + bit_of(w2, r.w) >= 2;
}

__host__ __device__ constexpr uint32_t flip3_hdr_words(uint32_t list_cap) { return 33 + list_cap / 4 + 1; }

// Set-up of one shot, by one whole wave: noise (or the caller's word), syndrome and flippable mask into the shot's
// LDS state S (Sw words, zeroed here together with the mask), F = S + Sw; the input word is parked at O (HBM) for the
// output stage.  `list` (LIST_CAP u16) and `cnt_lds` are this wave's own LDS.  Returns the summary of F (bit w =
// window w holds a flippable bit), the same value in every lane.  i = the shot's index within the call.
__device__ inline uint32_t flip3_setup_shot(const DeviceGraph &g, const Nb3Table &nbt, const FlipArgs &a, const uint32_t *q,
                                            uint16_t *list, uint32_t LIST_CAP, uint32_t *cnt_lds, uint64_t *S, uint64_t *O,
                                            uint64_t i, uint32_t lane, unsigned long long *stamp = nullptr)
{
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if (stamp) ts0 = __builtin_amdgcn_s_memtime();
    const uint32_t Wn = (g.n + 63) / 64, Wm = (g.m + 63) / 64;
    const uint32_t Sw = nbt.sw, per_shot = Sw + nbt.windows * 8;
    const uint4 *nb = nbt.rec;
    const uint32_t dummy_row = g.n;
    uint64_t *F = S + Sw;
    char *Sc = (char *)S, *Fc = (char *)F;
    const uint64_t shot = a.first_shot + i;
    uint32_t summary = 0;
    for (uint32_t w = lane; w < per_shot; w += 64) S[w] = 0;
    if (lane == 0) *cnt_lds = 0;
    wave_order();
    // noise (or the caller's word); error positions go to the LDS list
    if (a.source == SRC_BSC) {
        // A lane samples the words lane, lane + 64, ... (at most four: n + 512 <= 16384).  Block 0 of every word
        // first; the few words that need further Philox blocks (four or more errors in 64 bits: 4 % at p = 0.02)
        // are then finished together, lowest pending word of each lane per pass, instead of one sparse pass per
        // round of words.  The stream of a word is the sequential one (DESIGN.md "Noise specification").
        const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32), s_lo = (uint32_t)shot, s_hi = (uint32_t)(shot >> 32);
        const uint32_t rounds = (Wn + 63) / 64;
        uint64_t xs[4] = {0, 0, 0, 0};
        uint32_t ps[4] = {0, 0, 0, 0}, pend = 0;
        if (!a.gap.p_is_zero) {
#pragma unroll
            for (uint32_t r = 0; r < 4; r++)
                if (r < rounds) {
                    // (w is the same for every shot of this wave: left to the compiler, the first Philox product of each of
                    // the four words is computed once, spilled, and read back from scratch memory for every shot)
                    // (likewise the ten round keys of Philox, the same for every shot: they are built, kept in scalar
                    // registers beyond the budget and fetched lane by lane from a spill register, a vector instruction
                    // each; with the key made opaque they are scalar additions in place)
                    uint32_t w = lane + 64 * r, kk0 = k0, kk1 = k1;
                    asm volatile("" : "+v"(w), "+s"(kk0), "+s"(kk1));
                    if (w < Wn && !bsc_block(q, a.gap.inv_log2q, kk0, kk1, s_lo, s_hi, w, 0, bits_in_word(g.n, w), xs[r], ps[r])) pend |= 1u << r;
                }
            uint32_t tb = 1;
            while (__ballot(pend != 0)) {
                if (pend) {
                    const uint32_t r = (uint32_t)__builtin_ctz(pend);
                    uint64_t x = r == 0 ? xs[0] : (r == 1 ? xs[1] : (r == 2 ? xs[2] : xs[3]));
                    uint32_t pp = r == 0 ? ps[0] : (r == 1 ? ps[1] : (r == 2 ? ps[2] : ps[3]));
                    const uint32_t w = lane + 64 * r;
                    uint32_t kk0 = k0, kk1 = k1;
                    asm volatile("" : "+s"(kk0), "+s"(kk1));
                    const bool done = bsc_block(q, a.gap.inv_log2q, kk0, kk1, s_lo, s_hi, w, tb, bits_in_word(g.n, w), x, pp);
#pragma unroll
                    for (uint32_t u = 0; u < 4; u++)
                        if (r == u) {
                            xs[u] = x;
                            ps[u] = pp;
                        }
                    if (done) {
                        pend &= pend - 1;
                        tb = 1;
                    } else tb++;
                }
            }
        }
        uint32_t cc = 0;
#pragma unroll
        for (uint32_t r = 0; r < 4; r++)
            if (r < rounds) {
                const uint32_t w = lane + 64 * r;
                if (w < Wn) O[w] = xs[r];
                cc += (uint32_t)__builtin_popcountll(xs[r]);
            }
        if (cc) {
            uint32_t off = atomicAdd(cnt_lds, cc);
#pragma unroll
            for (uint32_t r = 0; r < 4; r++) {
                uint64_t x = xs[r];
                while (x) {
                    if (off < LIST_CAP) list[off] = (uint16_t)(((lane + 64 * r) << 6) + (uint32_t)__builtin_ctzll(x));
                    off++;
                    x &= x - 1;
                }
            }
        }
    } else {
        for (uint32_t w = lane; w < Wn; w += 64) {
            const uint32_t nb_bits = bits_in_word(g.n, w);
            uint64_t x = a.words_in[i * Wn + w];
            if (nb_bits < 64) x &= (1ull << nb_bits) - 1;
            O[w] = x;
            const uint32_t cc = (uint32_t)__builtin_popcountll(x);
            if (cc) {
                uint32_t off = atomicAdd(cnt_lds, cc);
                while (x) {
                    if (off < LIST_CAP) list[off] = (uint16_t)((w << 6) + (uint32_t)__builtin_ctzll(x));
                    off++;
                    x &= x - 1;
                }
            }
        }
    }
    // (LDS only: the list and its counter.  The parked word is read back by the overflow path below, which waits for
    // it, and by the output stage, behind the release at the end of the set-up.  A service wave runs on memory round
    // trips and on the LDS, not on instruction issue -- a quarter fewer instructions in pass 2 bought nothing, one
    // round trip less in the output stage 2.8 % of the whole kernel -- so nothing here waits for a store that nobody
    // is about to read.)
    wave_order();
    if (stamp) ts1 = __builtin_amdgcn_s_memtime();
    const uint32_t total = __builtin_amdgcn_readfirstlane(*cnt_lds);
    bool synd_written = false;
    if (total <= LIST_CAP) {
        // (pass 2's list padding is written here, ahead of pass 1, which only reads entries below `total`)
        constexpr uint32_t P2 = 8;
        const uint32_t padded = (total + 4 * P2 - 1) & ~(4 * P2 - 1);   // <= LIST_CAP (a multiple of 64)
        if (total + lane < padded) list[total + lane] = (uint16_t)dummy_row;
        wave_order();
        const char *nb_base = (const char *)nb;
        const uint32_t sl16 = (lane & 15) << 4;
        const uint16_t *lp = list + (lane >> 4);
        auto load = [&](uint4(&r)[P2], uint32_t e0) {
#pragma unroll
            for (uint32_t u = 0; u < P2; u++) r[u] = *(const uint4 *)(nb_base + (((uint32_t)lp[e0 + 4 * u] << 8) | sl16));
        };
        uint4 ra[P2], rb[P2];
        // pass 1, syndrome: toggle the three checks of every error bit (slot 0 of its row)
        for (uint32_t t0i = lane; t0i < total; t0i += 64 * 4) {
            uint4 r[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++)
                if (t0i + 64 * u < total) r[u] = nb[(size_t)list[t0i + 64 * u] << 4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++)
                if (t0i + 64 * u < total) {
                    atomicXor((uint32_t *)(Sc + (r[u].y >> 16)), 1u << (r[u].y & 31));
                    atomicXor((uint32_t *)(Sc + (r[u].z >> 16)), 1u << (r[u].z & 31));
                    atomicXor((uint32_t *)(Sc + (r[u].w >> 16)), 1u << (r[u].w & 31));
                }
        }
        wave_order();
        if (stamp) ts2 = __builtin_amdgcn_s_memtime();
        // (the syndrome is final: its record goes out now, and is long in memory when the set-up ends)
        if (a.synd_out)
            for (uint32_t w = lane; w < Wm; w += 64) a.synd_out[i * Wm + w] = S[w];
        synd_written = true;
        // pass 2, flippable mask: only bits next to an error bit can see an unsatisfied check; a padding
        // slot reads the spare check and is never flippable.  Lane = slot (lane & 15) of the errors lane / 16 + 4 u:
        // the list is padded with the spare row to a multiple of 32 entries, so no lane needs a bounds test, the list
        // reads differ by constant offsets and a row address is one shift-or on a 32-bit offset.
        // (two sets of records: the rows of the next 32 errors are on their way while this set is evaluated; a set's
        // 24 syndrome words are read together, then the slots decide)
        // (a flippable slot sets its mask bit under an exec mask made by the compare itself -- v_cmpx -- instead of a
        // branch that, with 64 slots per wave, was always taken; the summary of F is read off F once, after the pass)
        const uint32_t Fc_addr = lds_addr_of(Fc);
        auto eval = [&](const uint4(&r)[P2]) {
            uint32_t wa[P2], wb[P2], wc[P2];
#pragma unroll
            for (uint32_t u = 0; u < P2; u++) {
                wa[u] = *(const uint32_t *)(Sc + (r[u].y >> 16));
                wb[u] = *(const uint32_t *)(Sc + (r[u].z >> 16));
                wc[u] = *(const uint32_t *)(Sc + (r[u].w >> 16));
            }
#pragma unroll
            for (uint32_t u = 0; u < P2; u++) {
                const uint32_t t = bit_of(wa[u], r[u].y) + bit_of(wb[u], r[u].z) + bit_of(wc[u], r[u].w);   // flip.rs:41: t >= 2
                const uint32_t fa = Fc_addr + (r[u].x >> 16), fm = 1u << (r[u].x & 31);
                uint64_t sv;
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "v_cmpx_lt_u32 vcc, 1, %[t]\n\t"
                             "ds_or_b32 %[fa], %[fm]\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [sv] "=&s"(sv)
                             : [t] "v"(t), [fa] "v"(fa), [fm] "v"(fm)
                             : "memory", "vcc");
            }
        };
        // (asking for these rows before pass 1 works its own measured slower: 40.1 -> 40.6 ms per 2^22 shots)
        if (padded) load(ra, 0);
        for (uint32_t e0 = 0; e0 < padded; e0 += 8 * P2) {
            const bool second = e0 + 4 * P2 < padded;
            if (second) load(rb, e0 + 4 * P2);
            eval(ra);
            if (second) {
                if (e0 + 8 * P2 < padded) load(ra, e0 + 8 * P2);
                eval(rb);
            }
        }
    } else {
        // more errors than the list holds: the same two passes over pieces of LIST_CAP / 64 words
        wave_sync();   // (the parked word is read back from memory)
        const uint32_t pw = LIST_CAP / 64;
        for (int pass = 0; pass < 2; pass++) {
            for (uint32_t wb = 0; wb < Wn; wb += pw) {
                wave_order();
                if (lane == 0) *cnt_lds = 0;
                wave_order();
                const uint32_t w = wb + lane;
                uint64_t x = (lane < pw && w < Wn) ? O[w] : 0;
                const uint32_t cc = (uint32_t)__builtin_popcountll(x);
                if (cc) {
                    uint32_t off = atomicAdd(cnt_lds, cc);
                    while (x) {
                        list[off++] = (uint16_t)((w << 6) + (uint32_t)__builtin_ctzll(x));
                        x &= x - 1;
                    }
                }
                wave_order();
                const uint32_t cnt = __builtin_amdgcn_readfirstlane(*cnt_lds);
                if (pass == 0) {
                    for (uint32_t t = lane; t < cnt; t += 64) {
                        const uint4 r = nb[(size_t)list[t] << 4];
                        atomicXor((uint32_t *)(Sc + (r.y >> 16)), 1u << (r.y & 31));
                        atomicXor((uint32_t *)(Sc + (r.z >> 16)), 1u << (r.z & 31));
                        atomicXor((uint32_t *)(Sc + (r.w >> 16)), 1u << (r.w & 31));
                    }
                } else {
                    for (uint32_t t = lane; t < (cnt << 4); t += 64) {
                        const uint4 r = nb[((size_t)list[t >> 4] << 4) + (t & 15)];
                        if (rec3_flippable(r, Sc)) {
                            atomicOr((uint32_t *)(Fc + (r.x >> 16)), 1u << (r.x & 31));
                            summary |= 1u << (r.x >> 22);
                        }
                    }
                }
            }
            wave_order();
        }
    }
    wave_order();
    if (a.synd_out && !synd_written)
        for (uint32_t w = lane; w < Wm; w += 64) a.synd_out[i * Wm + w] = S[w];
    // summary of F: bit w = window w (16 mask words) holds a flippable bit
    {
        const uint32_t *F32 = (const uint32_t *)F;
        for (uint32_t d = lane; d < nbt.windows * 16u; d += 64)
            if (F32[d]) summary |= 1u << (d >> 4);
    }
    const uint32_t all = (uint32_t)wave_or64(summary);
    if (stamp) {   // diagnostics: noise + list, pass 1, pass 2 + syndrome output
        const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
        stamp[0] += ts1 - ts0; stamp[1] += ts2 - ts1; stamp[2] += ts3 - ts2;
    }
    return all;
}

// Output of one finished shot, by one whole wave: the decoded word = parked input word (E, HBM) with the nf logged
// flips (lj, HBM) applied (flip.rs:30, deferred), rebuilt in the shot's dead F region; failure test against the
// all-zero codeword.  Returns bit 0 = decoded word non-zero (failure), bit 1 = final syndrome non-zero (detected).
__device__ inline uint32_t flip3_output_shot(const DeviceGraph &g, const Nb3Table &nbt, const FlipArgs &a, uint64_t *S,
                                             const uint64_t *E, const uint16_t *lj, uint32_t nf, uint64_t i, uint32_t lane)
{
    const uint32_t Wn = (g.n + 63) / 64, Wm = (g.m + 63) / 64;
    uint64_t *Oj = S + nbt.sw;
    uint32_t *O32 = (uint32_t *)Oj;
    // device-scope loads: both were written through other lanes / waves of this CU.  The log is read two entries a load
    // (it is 32-byte aligned), and its first 512 entries are asked for together with the parked word: one memory round
    // trip for both.
    const uint32_t *lj32 = (const uint32_t *)lj;
    const uint32_t npair = (nf + 1) >> 1;
    uint32_t bb[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; u++)
        bb[u] = lane + 64 * u < npair ? __hip_atomic_load(&lj32[lane + 64 * u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    for (uint32_t w = lane; w < Wn; w += 64) Oj[w] = __hip_atomic_load(&E[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wave_order();
    for (uint32_t t0i = lane;;) {
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t pi = t0i + 64 * u;
            if (pi < npair) {
                const uint32_t b0 = bb[u] & 0xFFFFu, b1 = bb[u] >> 16;
                atomicXor(&O32[b0 >> 5], 1u << (b0 & 31));                                     // flip.rs:30
                if (2 * pi + 1 < nf) atomicXor(&O32[b1 >> 5], 1u << (b1 & 31));
            }
        }
        t0i += 64 * 4;
        if (t0i - lane >= npair) break;
#pragma unroll
        for (uint32_t u = 0; u < 4; u++)
            bb[u] = t0i + 64 * u < npair ? __hip_atomic_load(&lj32[t0i + 64 * u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    wave_order();
    uint64_t anyo = 0, anys = 0;
    for (uint32_t w = lane; w < Wn; w += 64) {
        const uint64_t x = Oj[w];
        anyo |= x;
        if (a.words_out) a.words_out[i * Wn + w] = x;
    }
    for (uint32_t w = lane; w < Wm; w += 64) anys |= S[w];
    const bool fail = __ballot(anyo != 0) != 0, det = __ballot(anys != 0) != 0;
    if (a.flips_out && lane == 0) a.flips_out[i] = nf;
    return (fail ? 1u : 0u) | (det ? 2u : 0u);
}

// One iteration of the serial flip loop for the eight shots of a wave (8 lanes each): the lowest flippable bit of
// each shot is flipped (flip.rs:27-30), or, when the scanned window of its mask holds none, the window is retired.
// Sg / Fg: the group's shot state in LDS; fs2: summary of F (0: nothing to do -- the group then works the spare row
// and changes nothing); nfl: flips so far = position in the flip log.
template <bool STAMP = false, int SV = 0>
__device__ inline void flip3_step(const char *nb_base, uint32_t dummy_row, char *Sg, char *Fg, char *log_base, uint32_t log_off,
                                  uint32_t gl, uint32_t &fs2, uint32_t &nfl, unsigned long long *stamp = nullptr)
{
    // STAMP (diagnostics): cycles of the four stretches of a pass -- mask window read, search, row load, evaluation
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (STAMP) t0 = __builtin_amdgcn_s_memtime();
    const uint32_t Fg_addr = lds_addr_of(Fg);
    const uint32_t w2 = fs2 ? (uint32_t)__builtin_ctz(fs2) : 0u;
    const uint64_t x = *(const uint64_t *)(Fg + (w2 << 6) + gl * 8);
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (STAMP) {
        asm volatile("" : "+v"(lo), "+v"(hi));
        t1 = __builtin_amdgcn_s_memtime();
    }
    // lowest set bit of the lane's 64 mask bits as a position in the window (v_ffbl_b32 gives all ones for zero, and
    // all ones stay all ones under the OR with the lane's base)
    uint32_t clo, chi;
    asm("v_ffbl_b32 %0, %1" : "=v"(clo) : "v"(lo));
    asm("v_ffbl_b32 %0, %1" : "=v"(chi) : "v"(hi));
    clo |= gl << 6;
    chi |= (gl << 6) | 32u;
    const uint32_t c = clo < chi ? clo : chi;
    const uint32_t mpos = half_row_min2_u32(c);
    const bool found = fs2 != 0 && mpos != NONE;
    // the window is empty, or the bit about to be flipped is its only one: retire it (a neighbour that
    // becomes flippable in it sets the summary bit again below)
    const uint32_t inwin = half_row_add_u32((uint32_t)__builtin_popcount(lo) + (uint32_t)__builtin_popcount(hi));
    if (inwin <= 1) fs2 &= fs2 - 1;
    const uint32_t b = found ? (w2 << 9) | mpos : dummy_row;
    if (STAMP) {
        uint32_t bb = b;
        asm volatile("" : "+v"(bb));
        t2 = __builtin_amdgcn_s_memtime();
    }
    // (uniform base + 32-bit lane offset: one address instruction per load; the table is at most 4 MiB)
    const uint32_t roff = (b << 8) + gl * 32u;
    uint4 r0 = *(const uint4 *)(nb_base + roff), r1 = *(const uint4 *)(nb_base + roff + 16);
    if (STAMP) {
        asm volatile("" : "+v"(r0.x), "+v"(r0.y), "+v"(r0.z), "+v"(r0.w), "+v"(r1.x), "+v"(r1.y), "+v"(r1.z), "+v"(r1.w));
        t3 = __builtin_amdgcn_s_memtime();
    }
    const uint32_t one = found ? 1u : 0u;
    if (gl == 0) {                                             // slot 0 is b itself: syndrome ^= H e_b (flip.rs:28-29)
        atomicXor((uint32_t *)(Sg + (r0.y >> 16)), one << (r0.y & 31));
        atomicXor((uint32_t *)(Sg + (r0.z >> 16)), one << (r0.z & 31));
        atomicXor((uint32_t *)(Sg + (r0.w >> 16)), one << (r0.w & 31));
        *(uint16_t *)(log_base + (log_off + nfl * 2u)) = (uint16_t)b;   // output ^= e_b (flip.rs:30) is deferred: log the flip
    }
    wave_order();
    // Both slots re-evaluate their bit against the new syndrome.  One of a neighbour's checks (word 3 of its record)
    // changed by this flip, so its mask bit changes only if the count now sits at the threshold on the side it came
    // from: count == 1 + that check, i.e. the other two checks hold exactly one unsatisfied; the record's flag covers
    // the slots for which that does not hold (b itself, doubly connected bits).  Few lanes write, so the mask update
    // costs the LDS little.
    // All six syndrome words are read before either slot decides: one LDS round trip on the pass's dependent chain
    // instead of two (+6 % at C2 under flip4_kernel).  Other forms of this stretch measured there: every lane
    // rewriting its bit, or writing under an empty mask instead of branching (-0.5 %); the third word read only by the
    // lanes that write (no difference); the three toggles spread over three lanes' records so that one LDS
    // instruction carries them (-1.7 %, although it saves nine instructions).  (SV: spare switch for such A/B forms,
    // LDPC_HIP_FLIP_NF = 100 * SV + flip waves.)
    uint32_t contrib = 0;
    uint32_t w6[6];
    w6[0] = *(const uint32_t *)(Sg + (r0.y >> 16)); w6[1] = *(const uint32_t *)(Sg + (r0.z >> 16)); w6[2] = *(const uint32_t *)(Sg + (r0.w >> 16));
    w6[3] = *(const uint32_t *)(Sg + (r1.y >> 16)); w6[4] = *(const uint32_t *)(Sg + (r1.z >> 16)); w6[5] = *(const uint32_t *)(Sg + (r1.w >> 16));
    asm volatile("" : "+v"(w6[0]), "+v"(w6[1]), "+v"(w6[2]), "+v"(w6[3]), "+v"(w6[4]), "+v"(w6[5]));
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const uint4 &r = s ? r1 : r0;
        const uint32_t ab = bit_of(w6[3 * s], r.y) + bit_of(w6[3 * s + 1], r.z);
        if ((r.x & 0x100u) || ab == 1u) {
            const bool fl = ab + bit_of(w6[3 * s + 2], r.w) >= 2;   // flip.rs:41 for three checks
            const uint32_t m = one << (r.x & 31);
            lds_mskor(Fg_addr + (r.x >> 16), m, fl ? m : 0u);
            if (fl) contrib |= 1u << (r.x >> 22);
        }
    }
    fs2 |= half_row_or_u32(contrib);
    nfl += one;
    // (issue-slot probes: what ten more instructions of either kind cost a pass, profiles/README.md round 3)
    if (SV == 1) asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
    if (SV == 2) {
        uint32_t t_ = gl;
        asm volatile("v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n"
                     "v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0" : "+v"(t_));
    }
    wave_order();
    if (STAMP) {
        uint32_t f = fs2;
        asm volatile("" : "+v"(f));
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        stamp[0] += t1 - t0; stamp[1] += t2 - t1; stamp[2] += t3 - t2; stamp[3] += t4 - t3;
    }
}

// 32-bit LDS loads at a byte address kept in a register (the same address feeds the asm blocks below)
__device__ inline uint32_t lds_ld32(uint32_t addr) { return *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)addr; }
__device__ inline uint64_t lds_ld64(uint32_t addr) { return *(const __attribute__((address_space(3))) uint64_t *)(uintptr_t)addr; }

// The pass of flip3_step with the wave's own instruction stream cut down (round 3: ten more instructions in a pass,
// scalar or vector, cost it 3.6 - 4.1 %, i.e. a pass is mostly the wave issuing its ~140 instructions at one per
// 4.5 - 5 cycles; profiles/README.md).  Same decisions, fewer instructions:
//   * no `found` predicate: an idle group reads a zeroed window (Fz_addr below names one), finds nothing, and the
//     minimum with the spare row's index turns "nothing" into the spare row; fs2 == 0 scans the LAST window, which
//     is empty like all the others then;
//   * the three syndrome toggles and the log entry of the group's first lane, and the two masked mask writes, run
//     under an exec mask set by hand (no branch around them: with eight shots per wave the branches were always taken);
//   * a slot is evaluated from all three of its checks at once, and everything that does not need the syndrome words
//     (mask word address, mask bit, window bit, the record's flag) is computed while they are on their way;
//   * no per-pass watchdog counter in vector registers (the caller bounds the loop with a scalar budget).
// Sg_addr / Fg_addr: LDS byte addresses of the group's syndrome and mask (Fg_addr: of a zeroed 64-byte block minus
// wlast * 64 for a group without a shot); wlast = windows - 1.
__device__ inline void flip3_step_v2(const char *nb_base, uint32_t dummy_row, uint32_t wlast, uint32_t Sg_addr, uint32_t Fg_addr,
                                     uint32_t Fg_lane, const char *log_base, uint32_t log_off, uint32_t gl, uint32_t &fs2, uint32_t &nfl)
{
    uint32_t w2;
    asm("v_ffbl_b32 %0, %1" : "=v"(w2) : "v"(fs2));
    w2 = w2 < wlast ? w2 : wlast;
    const uint64_t x = lds_ld64(Fg_lane + (w2 << 6));   // Fg_lane = Fg_addr + 8 gl
    const uint32_t fs2r = fs2 & (fs2 - 1);        // the summary with the scanned window retired
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    uint32_t clo, chi;
    asm("v_ffbl_b32 %0, %1" : "=v"(clo) : "v"(lo));
    asm("v_ffbl_b32 %0, %1" : "=v"(chi) : "v"(hi));
    clo |= gl << 6;
    chi |= (gl << 6) | 32u;
    const uint32_t c = clo < chi ? clo : chi;
    const uint32_t mpos = half_row_min2_u32(c);
    const uint32_t cand = (w2 << 9) | mpos;       // (all ones when the window is empty)
    const uint32_t b = cand < dummy_row ? cand : dummy_row;
    const uint32_t roff = (b << 8) + gl * 32u;
    const uint4 r0 = *(const uint4 *)(nb_base + roff), r1 = *(const uint4 *)(nb_base + roff + 16);
    // while the row is on its way: the window is empty, or the bit about to be flipped is its only one -> retire it
    const uint32_t inwin = half_row_add_u32((uint32_t)__builtin_popcount(lo) + (uint32_t)__builtin_popcount(hi));
    const uint32_t keep = inwin <= 1 ? fs2r : fs2;
    const uint32_t one = b < dummy_row ? 1u : 0u;
    const uint32_t log_at = log_off + nfl * 2u;
    nfl += one;
    // slot 0 of the group's first lane is b itself: syndrome ^= H e_b (flip.rs:28-29); output ^= e_b (flip.rs:30) is
    // deferred: the flip is logged
    const uint32_t a0 = Sg_addr + (r0.y >> 16), a1 = Sg_addr + (r0.z >> 16), a2 = Sg_addr + (r0.w >> 16);
    const uint32_t a3 = Sg_addr + (r1.y >> 16), a4 = Sg_addr + (r1.z >> 16), a5 = Sg_addr + (r1.w >> 16);
    {
        const uint32_t d0 = one << (r0.y & 31), d1 = one << (r0.z & 31), d2 = one << (r0.w & 31);
        uint64_t sv;
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "s_and_b64 exec, exec, %[l0]\n\t"
                     "ds_xor_b32 %[a0], %[d0]\n\t"
                     "ds_xor_b32 %[a1], %[d1]\n\t"
                     "ds_xor_b32 %[a2], %[d2]\n\t"
                     "global_store_short %[la], %[b], %[lb]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [sv] "=&s"(sv)
                     : [l0] "s"(0x0101010101010101ull), [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2),
                       [la] "v"(log_at), [b] "v"(b), [lb] "s"(log_base)
                     : "memory");
    }
    // both slots of the lane against the new syndrome: six words in one LDS round trip
    uint32_t w0 = lds_ld32(a0), w1 = lds_ld32(a1), w2s = lds_ld32(a2), w3 = lds_ld32(a3), w4 = lds_ld32(a4), w5 = lds_ld32(a5);
    // (on the way: where and what a slot would write)
    uint32_t fa0 = Fg_addr + (r0.x >> 16), fa1 = Fg_addr + (r1.x >> 16);
    uint32_t m0 = one << (r0.x & 31), m1 = one << (r1.x & 31);
    uint32_t wb0 = 1u << (r0.x >> 22), wb1 = 1u << (r1.x >> 22);
    uint64_t gm0, gm1;   // lane masks: the record is flagged
    {
        const uint32_t g0 = r0.x & 0x100u, g1 = r1.x & 0x100u;
        asm("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(gm0) : "v"(g0));
        asm("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(gm1) : "v"(g1));
    }
    asm volatile("" : "+v"(fa0), "+v"(fa1), "+v"(m0), "+v"(m1), "+v"(wb0), "+v"(wb1), "+s"(gm0), "+s"(gm1));
    asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2s), "+v"(w3), "+v"(w4), "+v"(w5));
    // One of a neighbour's checks (word 3 of its record) changed by this flip, so its mask bit changes only if the
    // other two hold exactly one unsatisfied check; the record's flag covers the slots for which that does not hold
    // (b itself, doubly connected bits).  flip.rs:41 for three checks: flippable iff two or more are unsatisfied.
    const uint32_t x0 = bit_of(w0, r0.y), y0 = bit_of(w1, r0.z), t0 = x0 + y0 + bit_of(w2s, r0.w);
    const uint32_t x1 = bit_of(w3, r1.y), y1 = bit_of(w4, r1.z), t1 = x1 + y1 + bit_of(w5, r1.w);
    const uint32_t v0 = t0 >= 2 ? m0 : 0u, v1 = t1 >= 2 ? m1 : 0u;
    {
        uint64_t sv, c0, c1;
        asm volatile("v_cmp_ne_u32_e64 %[c0], %[x0], %[y0]\n\t"
                     "v_cmp_ne_u32_e64 %[c1], %[x1], %[y1]\n\t"
                     "s_mov_b64 %[sv], exec\n\t"
                     "s_or_b64 %[c0], %[c0], %[g0]\n\t"
                     "s_or_b64 %[c1], %[c1], %[g1]\n\t"
                     "s_and_b64 exec, %[sv], %[c0]\n\t"
                     "ds_mskor_b32 %[fa0], %[m0], %[v0]\n\t"
                     "s_and_b64 exec, %[sv], %[c1]\n\t"
                     "ds_mskor_b32 %[fa1], %[m1], %[v1]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [sv] "=&s"(sv), [c0] "=&s"(c0), [c1] "=&s"(c1)
                     : [g0] "s"(gm0), [g1] "s"(gm1), [x0] "v"(x0), [y0] "v"(y0), [x1] "v"(x1), [y1] "v"(y1), [fa0] "v"(fa0), [fa1] "v"(fa1),
                       [m0] "v"(m0), [m1] "v"(m1), [v0] "v"(v0), [v1] "v"(v1)
                     : "memory");
    }
    const uint32_t contrib = (t0 >= 2 ? wb0 : 0u) | (t1 >= 2 ? wb1 : 0u);
    fs2 = keep | half_row_or_u32(contrib);
}

// The loop of passes of a flip wave, written out by hand (same pass as flip3_step_v2, which is its readable form and
// stays as the A/B reference): runs passes until some group of the wave has nothing left to flip (fs2 == 0 in its
// lanes) or `left` passes have run (watchdog: left comes back negative).  What the hand-written form adds to
// flip3_step_v2: the two DPP reductions of the search interleaved with each other and with the pass's bookkeeping so
// that no DPP hazard costs a no-op, exec set straight from the compare results (the wave runs with all 64 lanes, which
// flip4_kernel guarantees), loop control of three scalar instructions, and the last reduction step writing fs2.
// 95 instructions per pass (66 vector, 12 LDS, 3 vector-memory) against 117 compiled from flip3_step_v2 and 140 from
// flip3_step; temporaries are fixed registers v80-v124, s[60:73] (clobbered).
__device__ inline void flip3_passes_asm(const char *nb_base, uint32_t dummy_row, uint32_t wlast, uint32_t Sg_addr, uint32_t Fg_addr,
                                        uint32_t Fg_lane, const char *log_base, uint32_t log_off, uint32_t gl, uint32_t &fs2,
                                        uint32_t &nfl, int32_t &left)
{
    const uint32_t k1 = gl << 6, k2 = (gl << 6) | 32u, k3 = gl * 32u;
    asm volatile(
        "s_mov_b64 s[68:69], exec\n\t"
        "s_and_b64 s[68:69], s[68:69], %[l0]\n"                    // first lane of every group of eight
        "1:\n\t"
        "v_ffbl_b32 v80, %[fs2]\n\t"
        "v_min_u32 v80, %[wl], v80\n\t"                            // window to scan (fs2 == 0: the last one, empty)
        "v_lshl_add_u32 v81, v80, 6, %[Fl]\n\t"
        "ds_read_b64 v[82:83], v81\n\t"
        "v_lshl_add_u32 v85, %[nfl], 1, %[lo]\n\t"                 // (on the way) where the flip is logged
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_ffbl_b32 v86, v82\n\t"
        "v_ffbl_b32 v87, v83\n\t"
        "v_or_b32 v86, v86, %[k1]\n\t"
        "v_or_b32 v87, v87, %[k2]\n\t"
        "v_min_u32 v86, v86, v87\n\t"
        "v_bcnt_u32_b32 v88, v82, 0\n\t"
        "v_bcnt_u32_b32 v88, v83, v88\n\t"
        "v_min_u32_dpp v86, v86, v86 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_u32 v84, -1, %[fs2]\n\t"
        "v_add_u32_dpp v88, v88, v88 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_min_u32_dpp v86, v86, v86 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_and_b32 v84, v84, %[fs2]\n\t"                           // the summary with the scanned window retired
        "v_add_u32_dpp v88, v88, v88 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_min_u32_dpp v86, v86, v86 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_lshl_or_b32 v86, v80, 9, v86\n\t"                       // lowest flippable bit (all ones: none)
        "v_min_u32 v89, %[dr], v86\n\t"                            // ... or the spare row
        "v_lshl_or_b32 v90, v89, 8, %[k3]\n\t"
        "global_load_dwordx4 v[92:95], v90, %[nb]\n\t"
        "global_load_dwordx4 v[96:99], v90, %[nb] offset:16\n\t"
        "v_add_u32_dpp v88, v88, v88 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        // while the row is on its way: retire the window if the bit about to be flipped was its only one; count the flip
        // (gfx950: a lane mask written by a vector compare may be read by a vector instruction two instructions later at
        // the earliest -- the compiler pads with s_nop; here the order provides the distance, and this stretch is in the
        // shadow of the row load anyway)
        "v_cmp_gt_u32_e64 s[60:61], 2, v88\n\t"
        "v_cmp_gt_u32_e64 s[62:63], %[dr], v86\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 v84, %[fs2], v84, s[60:61]\n\t"
        "v_cndmask_b32_e64 v91, 0, 1, s[62:63]\n\t"
        "v_add_u32 %[nfl], %[nfl], v91\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        // slot 0 of the group's first lane is the bit itself: syndrome ^= H e_b (flip.rs:28-29), the flip is logged
        "v_add_u32_sdwa v100, v93, %[Sg] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_lshlrev_b32 v106, v93, v91\n\t"
        "v_add_u32_sdwa v101, v94, %[Sg] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_lshlrev_b32 v107, v94, v91\n\t"
        "v_add_u32_sdwa v102, v95, %[Sg] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_lshlrev_b32 v108, v95, v91\n\t"
        "s_mov_b64 exec, s[68:69]\n\t"
        "ds_xor_b32 v100, v106\n\t"
        "ds_xor_b32 v101, v107\n\t"
        "ds_xor_b32 v102, v108\n\t"
        "global_store_short v85, v89, %[lb]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_waitcnt vmcnt(1)\n\t"                                   // (the second half of the row; the store stays out)
        "v_add_u32_sdwa v103, v97, %[Sg] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_add_u32_sdwa v104, v98, %[Sg] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_add_u32_sdwa v105, v99, %[Sg] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
        // both slots of the lane against the new syndrome: six words in one LDS round trip
        "ds_read_b32 v109, v100\n\t"
        "ds_read_b32 v110, v101\n\t"
        "ds_read_b32 v111, v102\n\t"
        "ds_read_b32 v112, v103\n\t"
        "ds_read_b32 v113, v104\n\t"
        "ds_read_b32 v114, v105\n\t"
#ifdef XP_LDS_PROBE   // (what two more random 64-lane LDS reads cost a pass: profiles/README.md round 3)
        "ds_read_b32 v125, v103\n\t"
        "ds_read_b32 v126, v104\n\t"
#endif
        // on the way: where and what a slot would write, its window, its flag
        "v_add_u32_sdwa v115, v92, %[Fg] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_add_u32_sdwa v116, v96, %[Fg] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_lshlrev_b32 v117, v92, v91\n\t"
        "v_lshlrev_b32 v118, v96, v91\n\t"
        "v_lshrrev_b32 v119, 22, v92\n\t"
        "v_lshrrev_b32 v120, 22, v96\n\t"
        "v_and_b32 v121, 0x100, v92\n\t"
        "v_and_b32 v122, 0x100, v96\n\t"
        "v_lshlrev_b32 v119, v119, 1\n\t"
        "v_lshlrev_b32 v120, v120, 1\n\t"
        "v_cmp_ne_u32_e64 s[60:61], 0, v121\n\t"
        "v_cmp_ne_u32_e64 s[62:63], 0, v122\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        // One of a neighbour's checks changed by this flip, so its mask bit changes only if the other two hold exactly
        // one unsatisfied check; the flag covers the records for which that does not hold.  flip.rs:41 for three
        // checks: flippable iff two or more are unsatisfied.
        "v_bfe_u32 v109, v109, v93, 1\n\t"
        "v_bfe_u32 v110, v110, v94, 1\n\t"
        "v_bfe_u32 v111, v111, v95, 1\n\t"
        "v_bfe_u32 v112, v112, v97, 1\n\t"
        "v_bfe_u32 v113, v113, v98, 1\n\t"
        "v_bfe_u32 v114, v114, v99, 1\n\t"
        "v_add3_u32 v121, v109, v110, v111\n\t"
        "v_add3_u32 v122, v112, v113, v114\n\t"
        "v_cmp_lt_u32_e64 s[70:71], 1, v121\n\t"
        "v_cmp_lt_u32_e64 s[72:73], 1, v122\n\t"
        "v_cmp_ne_u32_e64 s[64:65], v109, v110\n\t"
        "v_cmp_ne_u32_e64 s[66:67], v112, v113\n\t"
        "v_cndmask_b32_e64 v123, 0, v117, s[70:71]\n\t"
        "v_cndmask_b32_e64 v119, 0, v119, s[70:71]\n\t"
        "v_cndmask_b32_e64 v124, 0, v118, s[72:73]\n\t"
        "v_cndmask_b32_e64 v120, 0, v120, s[72:73]\n\t"
        "s_or_b64 exec, s[64:65], s[60:61]\n\t"
        "ds_mskor_b32 v115, v117, v123\n\t"
        "s_or_b64 exec, s[66:67], s[62:63]\n\t"
        "ds_mskor_b32 v116, v118, v124\n\t"
        "s_mov_b64 exec, -1\n\t"
        "v_or3_b32 v119, v119, v120, v84\n\t"                      // windows that hold a flippable bit now, and the ones kept
        "s_sub_u32 %[left], %[left], 1\n\t"
        "s_nop 0\n\t"
        "v_or_b32_dpp v119, v119, v119 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_or_b32_dpp v119, v119, v119 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_or_b32_dpp %[fs2], v119, v119 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cmp_eq_u32 vcc, 0, %[fs2]\n\t"
        "s_cbranch_vccnz 2f\n\t"                                   // some group is out of flips
        "s_cbranch_scc0 1b\n"
        "2:"
        : [fs2] "+v"(fs2), [nfl] "+v"(nfl), [left] "+s"(left)
        : [wl] "s"(wlast), [dr] "s"(dummy_row), [Fl] "v"(Fg_lane), [Fg] "v"(Fg_addr), [Sg] "v"(Sg_addr), [lo] "v"(log_off), [k1] "v"(k1),
          [k2] "v"(k2), [k3] "v"(k3), [nb] "s"(nb_base), [lb] "s"(log_base), [l0] "s"(0x0101010101010101ull)
        : "memory", "vcc", "scc", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "v80", "v81", "v82", "v83", "v84",
          "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102",
          "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118",
          "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126");
}

template <bool PROF> __global__ __launch_bounds__(64) void flip3_kernel(DeviceGraph g, Nb3Table nbt, FlipArgs a)
{
    extern __shared__ uint64_t lds[];
    constexpr uint32_t SPW = 8;                      // shots per wave, 8 lanes each
    const uint32_t lane = threadIdx.x, grp = lane >> 3, gl = lane & 7;
    const uint32_t Wn = (g.n + 63) / 64;
    const uint32_t Sw = nbt.sw, windows = nbt.windows, Fw = windows * 8;
    const uint32_t per_shot = Sw + Fw;               // LDS words per shot: S (with the spare check), F in whole windows
    uint32_t *q = (uint32_t *)lds;
    uint16_t *list = (uint16_t *)(lds + 33);         // error positions of the shot being set up
    const uint32_t LIST_CAP = a.list_cap;
    uint32_t *cnt_lds = (uint32_t *)(lds + 33 + LIST_CAP / 4);
    uint64_t *state = lds + flip3_hdr_words(LIST_CAP);

    for (uint32_t i = lane; i < 66; i += 64) q[i] = i < 65 ? a.gap.q[i] : 0u;   // (q[65] = 0: bsc_block reads one past the table)
    wave_sync();

    uint64_t c_shots = 0, c_fail = 0, c_det = 0, c_undet = 0, c_flips = 0;
    unsigned long long t_setup = 0, t_loop = 0, t_out = 0, n_iter = 0;

    for (uint64_t base = (uint64_t)blockIdx.x * SPW; base < a.n_shots; base += (uint64_t)gridDim.x * SPW) {
        const uint32_t nvalid = a.n_shots - base < SPW ? (uint32_t)(a.n_shots - base) : SPW;
        unsigned long long t0 = 0;
        if (PROF) t0 = __builtin_amdgcn_s_memtime();
        // the sampled / given word of shot j is parked in HBM during the flip loop (the caller's record
        // buffer, or this block's scratch slice); the flips are logged and applied to it at the end
        auto err_words = [&](uint32_t j) -> uint64_t * {
            return a.errors_out ? a.errors_out + (base + j) * Wn : a.o_scratch + ((size_t)blockIdx.x * SPW + j) * Wn;
        };
        // ------------------------------------------------------------------ set-up, wave-wide, shot by shot
        uint32_t fs2 = 0;
        for (uint32_t j = 0; j < nvalid; j++) {
            const uint32_t s = flip3_setup_shot(g, nbt, a, q, list, LIST_CAP, cnt_lds, state + (size_t)j * per_shot, err_words(j),
                                                base + j, lane);
            if (grp == j) fs2 = s;
        }
        unsigned long long t1 = 0;
        if (PROF) t1 = __builtin_amdgcn_s_memtime();
        // ------------------------------------------------------------------ serial flips, 8 shots in lockstep
        uint32_t nfl = 0;
        {
            char *Sg = (char *)(state + (size_t)grp * per_shot), *Fg = Sg + (size_t)Sw * 8;
            // flip log of the group's shot (u16 entries: n < 2^16), as a byte offset from the block's first log
            char *log_base = (char *)((uint16_t *)a.flip_log + (size_t)blockIdx.x * SPW * a.flip_log_cap);
            const uint32_t log_off = grp * a.flip_log_cap * 2u;
            // every iteration flips a bit (at most m flips: each lowers the syndrome weight) or retires a window
            const uint32_t iter_cap = 18 * g.m + 64;
            for (uint32_t it = 0; it < iter_cap; it++) {
                if (!__ballot(fs2 != 0)) break;                            // flip.rs:27 find_flippable == None, all shots
                flip3_step((const char *)nbt.rec, g.n, Sg, Fg, log_base, log_off, gl, fs2, nfl);
                if (PROF) n_iter++;
            }
        }
        wave_sync();   // log entries are in memory
        unsigned long long t2 = 0;
        if (PROF) t2 = __builtin_amdgcn_s_memtime();
        // ---------------------------------------------------------------- failure test + outputs
        for (uint32_t j = 0; j < nvalid; j++) {
            const uint32_t nfj = __shfl(nfl, j * 8, 64), nf = nfj < a.flip_log_cap ? nfj : a.flip_log_cap;
            const uint32_t r = flip3_output_shot(g, nbt, a, state + (size_t)j * per_shot, err_words(j),
                                                 (const uint16_t *)a.flip_log + ((size_t)blockIdx.x * SPW + j) * a.flip_log_cap, nf,
                                                 base + j, lane);
            c_shots++; c_fail += r & 1; c_det += (r >> 1) & 1; c_undet += r == 1; c_flips += nf;
        }
        wave_sync();
        if (PROF) {
            const unsigned long long t3 = __builtin_amdgcn_s_memtime();
            t_setup += t1 - t0; t_loop += t2 - t1; t_out += t3 - t2;
        }
    }
    if (lane == 0 && a.counters) {
        atomicAdd(&a.counters[CNT_SHOTS], (unsigned long long)c_shots);
        if (c_fail) atomicAdd(&a.counters[CNT_FAIL], (unsigned long long)c_fail);
        if (c_det) atomicAdd(&a.counters[CNT_DET], (unsigned long long)c_det);
        if (c_undet) atomicAdd(&a.counters[CNT_UNDET], (unsigned long long)c_undet);
        if (c_flips) atomicAdd(&a.counters[CNT_WORK], (unsigned long long)c_flips);
        if (PROF) {   // diagnostics (LDPC_HIP_FLIP_PROF): wave iterations, cycles in the loop, cycles in set-up + output
            atomicAdd(&a.counters[CNT_CAPPED], n_iter);
            atomicAdd(&a.counters[CNT_XFAIL], t_loop);
            atomicAdd(&a.counters[CNT_ZFAIL], t_setup + t_out);
        }
    }
}

// ---------------------------------------------------------------------------
// Flip decoder, fourth form = the third with the work of a shot split over two
// kinds of waves of one 1024-thread workgroup per CU (large runs):
//   * NF flip waves run flip3_step and nothing else: eight shots each, and a
//     group of 8 lanes whose shot has finished hands it back and takes the next
//     prepared one, so all eight chains of a wave stay busy (no waiting for the
//     slowest of eight, no set-up in front of the loop);
//   * 16 - NF service waves prepare shots (noise, syndrome, flippable mask:
//     flip3_setup_shot) into free LDS slots and turn finished ones into outputs
//     (flip3_output_shot).
// A slot = one shot's state in LDS; its status word walks FREE -> (service
// wave) SETUP -> READY -> (flip group) RUNNING -> DONE -> (service wave) ... .
// All hand-offs are inside one workgroup: LDS status words written after a
// workgroup-scope release, taken with LDS compare-and-swap.
// ---------------------------------------------------------------------------
constexpr uint32_t F4_FREE = 0, F4_HELD = 1, F4_READY = 2, F4_RUNNING = 3, F4_DONE = 4;
constexpr uint32_t F4_MAX_SLOTS = 112;
constexpr uint32_t F4_LOGS = F4_MAX_SLOTS + 16;   // flip logs / parked words per block in HBM scratch

struct Flip4Ctl {
    uint32_t status[F4_MAX_SLOTS];
    uint32_t shot[F4_MAX_SLOTS];     // index of the slot's shot within this block's range
    uint32_t nfl[F4_MAX_SLOTS];      // flips logged (written by the flip group on DONE)
    uint32_t fs2[F4_MAX_SLOTS];      // summary of F (written by the service wave on READY)
    uint32_t next_shot;              // shots handed to service waves so far
    uint32_t n_flipped;              // shots whose flip loop has ended
    uint32_t n_output;               // shots written out
    uint32_t abort;                  // watchdog: a wave gave up waiting (never in a correct run)
    uint32_t cnt[16];                // per service wave: list fill counter
    uint32_t zero[16];               // an empty mask window: what a flip group without a shot scans (flip3_step_v2)
};

__host__ __device__ constexpr uint32_t flip4_hdr_words(uint32_t list_cap, uint32_t helpers)
{
    return 34 + (uint32_t)(sizeof(Flip4Ctl) / 8) + helpers * (list_cap / 4);
}

// G = lanes per shot in the flip waves (8: two slots per lane, flip3_step); NF + NH = waves of the workgroup.
// Forms with 16 lanes per shot (one slot per lane, two workgroups per CU) and with 4 (four slots per lane, 16 shots
// per wave) were built on the same frame and measured slower (6.8e7 and 6.2e7 against 7.8e7 / 8.7e7 shots/s at C2):
// sixteen lanes cost 1.6x the instructions per flip on pipes that are 80 % busy, four make the wave's own instruction
// stream of a pass 1.5x as long.
template <int NF, bool PROF = false, int G = 8, int NH = 16 - NF, int SV = 0>
__global__ __launch_bounds__((NF + NH) * 64) void flip4_kernel(DeviceGraph g, Nb3Table nbt, FlipArgs a, uint32_t n_slots)
{
    extern __shared__ uint64_t lds[];
    constexpr uint32_t GPW = 64 / G;   // shots per flip wave
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t grp = lane / G, gl = lane % G;
    const uint32_t Wn = (g.n + 63) / 64;
    const uint32_t Sw = nbt.sw, per_shot = Sw + nbt.windows * 8;
    const uint32_t LIST_CAP = a.list_cap;
    uint32_t *q = (uint32_t *)lds;
    Flip4Ctl *ctl = (Flip4Ctl *)(lds + 34);
    uint16_t *lists = (uint16_t *)(lds + 34 + sizeof(Flip4Ctl) / 8);
    uint64_t *state = lds + flip4_hdr_words(LIST_CAP, NH);
    // this block's contiguous range of the call's shots
    const uint64_t chunk = (a.n_shots + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * chunk;
    const uint32_t count = lo >= a.n_shots ? 0u : (uint32_t)(a.n_shots - lo < chunk ? a.n_shots - lo : chunk);
    if (count == 0) return;

    for (uint32_t i = threadIdx.x; i < 66; i += blockDim.x) q[i] = i < 65 ? a.gap.q[i] : 0u;   // (q[65] = 0: bsc_block reads one past the table)
    for (uint32_t i = threadIdx.x; i < sizeof(Flip4Ctl) / 4; i += blockDim.x) ((uint32_t *)ctl)[i] = 0;
    __syncthreads();

    // scratch in HBM, one piece per (block, slot): parked input word (unless the caller records errors), flip log
    auto err_words = [&](uint32_t slot, uint32_t rel) -> uint64_t * {
        return a.errors_out ? a.errors_out + (lo + rel) * Wn : a.o_scratch + ((size_t)blockIdx.x * F4_LOGS + slot) * Wn;
    };
    // (logs F4_MAX_SLOTS + w: waste bin of flip wave w, for the no-op rounds of groups without a shot)
    auto log_of = [&](uint32_t slot) -> uint16_t * {
        return (uint16_t *)a.flip_log + ((size_t)blockIdx.x * F4_LOGS + slot) * a.flip_log_cap;
    };
    // first slot with status `want` among slots [0, n_slots), or NONE (wave-uniform)
    auto find_slot = [&](uint32_t want) -> uint32_t {
        const uint64_t m0 = __ballot(lane < n_slots && __hip_atomic_load(&ctl->status[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == want);
        if (m0) return (uint32_t)__builtin_ctzll(m0);
        const uint64_t m1 = __ballot(lane + 64 < n_slots && __hip_atomic_load(&ctl->status[lane + 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == want);
        return m1 ? 64u + (uint32_t)__builtin_ctzll(m1) : NONE;
    };
    constexpr uint32_t IDLE_CAP = 1u << 22;   // polls without progress before a wave gives up (~ seconds)

    if (wave >= NF) {
        // ================================================================= service wave
        const uint32_t h = wave - NF;
        // (A raised priority for these waves, s_setprio 2, measured 18 % slower in round 3: they wait on memory, and the
        // flip waves they then push aside are the ones whose finished shots they need.  profiles/README.md.)
        uint16_t *list = lists + (size_t)h * LIST_CAP;
        uint32_t *cnt_lds = &ctl->cnt[h];
        uint64_t c_shots = 0, c_fail = 0, c_det = 0, c_undet = 0, c_flips = 0;
        uint32_t idle = 0;
        unsigned long long p_t0 = 0, p_busy = 0, p_start = 0, p_svc[4] = {0, 0, 0, 0};
        if (PROF) p_start = __builtin_amdgcn_s_memtime();
        for (;;) {
            if (PROF && p_t0) p_busy += __builtin_amdgcn_s_memtime() - p_t0;
            p_t0 = 0;
            if (__hip_atomic_load(&ctl->n_output, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= count) break;
            if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
            // a finished shot first (its slot is then reused), else a free slot while shots remain
            uint32_t slot = find_slot(F4_DONE), from = F4_DONE;
            if (slot == NONE && __hip_atomic_load(&ctl->next_shot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < count) {
                slot = find_slot(F4_FREE);
                from = F4_FREE;
            }
            uint32_t got = 0;
            if (slot != NONE) {
                if (lane == 0) {
                    uint32_t expect = from;
                    got = __hip_atomic_compare_exchange_strong(&ctl->status[slot], &expect, F4_HELD, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
                }
                got = __builtin_amdgcn_readfirstlane(got);
            }
            if (!got) {
                if (++idle > IDLE_CAP) {
                    if (lane == 0) __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                continue;
            }
            idle = 0;
            if (PROF) p_t0 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint64_t *S = state + (size_t)slot * per_shot;
            if (from == F4_DONE) {
                unsigned long long to0 = 0;
                if (PROF) to0 = __builtin_amdgcn_s_memtime();
                const uint32_t rel = __builtin_amdgcn_readfirstlane(ctl->shot[slot]);
                const uint32_t nfj = __builtin_amdgcn_readfirstlane(ctl->nfl[slot]), nf = nfj < a.flip_log_cap ? nfj : a.flip_log_cap;
                const uint32_t r = flip3_output_shot(g, nbt, a, S, err_words(slot, rel), log_of(slot), nf, lo + rel, lane);
                c_shots++; c_fail += r & 1; c_det += (r >> 1) & 1; c_undet += r == 1; c_flips += nf;
                wave_order();   // (the slot's LDS is read; its records in memory are nobody's to read inside the kernel)
                if (lane == 0) __hip_atomic_fetch_add(&ctl->n_output, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (PROF) p_svc[3] += __builtin_amdgcn_s_memtime() - to0;
            }
            // the next shot of the block goes into the slot, if there is one
            uint32_t rel = 0;
            if (lane == 0) rel = __hip_atomic_fetch_add(&ctl->next_shot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            rel = __builtin_amdgcn_readfirstlane(rel);
            if (rel >= count) {
                // (next_shot may run past count by the number of service waves; the slot is retired)
                if (lane == 0) __hip_atomic_store(&ctl->status[slot], F4_HELD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                continue;
            }
            const uint32_t s = flip3_setup_shot(g, nbt, a, q, list, LIST_CAP, cnt_lds, S, err_words(slot, rel), lo + rel, lane,
                                                PROF ? p_svc : nullptr);
            if (lane == 0) {
                ctl->shot[slot] = rel;
                ctl->fs2[slot] = s;
            }
            wave_sync();   // state, parked word: written before the slot is offered
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(&ctl->status[slot], F4_READY, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (lane == 0 && a.counters) {
            if (c_shots) atomicAdd(&a.counters[CNT_SHOTS], (unsigned long long)c_shots);
            if (c_fail) atomicAdd(&a.counters[CNT_FAIL], (unsigned long long)c_fail);
            if (c_det) atomicAdd(&a.counters[CNT_DET], (unsigned long long)c_det);
            if (c_undet) atomicAdd(&a.counters[CNT_UNDET], (unsigned long long)c_undet);
            if (c_flips) atomicAdd(&a.counters[CNT_WORK], (unsigned long long)c_flips);
            // diagnostics (LDPC_HIP_FLIP_PROF): service waves' busy ticks in the low half of a spare counter, their
            // total ticks in the high half (units of 2^10 ticks)
            if (PROF) atomicAdd(&a.counters[CNT_ZFAIL], (p_busy >> 10) | ((__builtin_amdgcn_s_memtime() - p_start) >> 10) << 32);
            if (PROF && blockIdx.x == 0 && h == 0 && c_shots)
                printf("[flip4 prof] service wave, ticks per shot: noise + list %.0f, pass 1 %.0f, pass 2 %.0f, output %.0f\n",
                       (double)p_svc[0] / c_shots, (double)p_svc[1] / c_shots, (double)p_svc[2] / c_shots, (double)p_svc[3] / c_shots);
        }
        return;
    }
    // ===================================================================== flip wave
    if (a.max_flips == 1) __builtin_amdgcn_s_setprio(3);   // (experiment) the serial chains win the issue arbitration
    constexpr bool V2 = SV == 3 || SV == 4;
    uint32_t slot = NONE, fs2 = 0, nfl = 0, idle = 0, steps = 0;
    char *Sg = (char *)state, *Fg = Sg + (size_t)Sw * 8;
    // (V2: LDS byte addresses; a group without a shot scans the zeroed window of the control block)
    const uint32_t wlast = nbt.windows - 1, Fz_addr = lds_addr_of(ctl->zero) - wlast * 64u;
    uint32_t Sg_addr = lds_addr_of(state), Fg_addr = Fz_addr;
    // flip logs as byte offsets from the block's first one (uniform base + 32-bit offset: F4_LOGS logs of < 2^17 bytes)
    char *const log_base = (char *)log_of(0);
    const uint32_t waste = (F4_MAX_SLOTS + wave) * a.flip_log_cap * 2u;
    uint32_t log_off = waste;
    const uint32_t step_cap = 18 * g.m + 64;   // flips (<= m, each lowers the syndrome weight) + window retirements of one shot
    unsigned long long p_iter = 0, p_live = 0, p_start = 0, p_stamp[4] = {0, 0, 0, 0};
    if (PROF) p_start = __builtin_amdgcn_s_memtime();
    for (;;) {
        if (__ballot(fs2 == 0)) {
            // ---- some group has nothing to flip: hand finished shots back, take prepared ones
            const bool finished = fs2 == 0 && slot != NONE;
            if (__ballot(finished)) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (log entries stored from inline asm: the compiler does not count them)
                wave_sync();   // this wave's log entries are in memory
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (finished && gl == 0) {
                    ctl->nfl[slot] = nfl;
                    __hip_atomic_store(&ctl->status[slot], F4_DONE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&ctl->n_flipped, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (finished) {
                    // the group's no-op rounds must not touch the slot's log (its next shot writes there)
                    slot = NONE;
                    log_off = waste;
                    nfl = 0;
                    Fg_addr = Fz_addr;
                }
            }
            // ready slots (two status words per lane), dealt to the groups that need one in lane order
            const uint64_t r0 = __ballot(lane < n_slots && __hip_atomic_load(&ctl->status[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == F4_READY);
            const uint64_t r1 = __ballot(lane + 64 < n_slots && __hip_atomic_load(&ctl->status[lane + 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == F4_READY);
            const uint64_t needm = __ballot(slot == NONE && gl == 0);
            bool took = false;
            if (r0 | r1) {
                // group k of the needy ones (in lane order) tries the k-th ready slot
                const uint32_t rank = (uint32_t)__builtin_popcountll(needm & ((1ull << (grp * G)) - 1));
                uint32_t cand = NONE;
                {
                    uint64_t x0 = r0, x1 = r1;
                    uint32_t k = rank;
                    // k-th set bit of (r1:r0); at most GPW steps
                    for (uint32_t s = 0; s < GPW; s++) {
                        if (s == k) cand = x0 ? (uint32_t)__builtin_ctzll(x0) : x1 ? 64u + (uint32_t)__builtin_ctzll(x1) : NONE;
                        if (x0) x0 &= x0 - 1;
                        else x1 &= x1 - 1;
                    }
                }
                uint32_t ok = 0;
                if (slot == NONE && cand != NONE && gl == 0) {
                    uint32_t expect = F4_READY;
                    ok = __hip_atomic_compare_exchange_strong(&ctl->status[cand], &expect, F4_RUNNING, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
                }
                ok = __shfl(ok, grp * G, 64);   // the group's lane 0 made the attempt
                cand = __shfl(cand, grp * G, 64);
                if (slot == NONE && ok) {
                    slot = cand;
                    took = true;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (took) {
                    fs2 = ctl->fs2[slot];
                    nfl = 0;
                    Sg = (char *)(state + (size_t)slot * per_shot);
                    Fg = Sg + (size_t)Sw * 8;
                    Sg_addr = lds_addr_of(Sg);
                    Fg_addr = lds_addr_of(Fg);
                    log_off = slot * a.flip_log_cap * 2u;
                    steps = 0;
                    // (a shot with nothing to flip has fs2 == 0 and is handed back on the next pass)
                }
            }
            if (!__ballot(slot != NONE)) {
                // every group is idle: done when the block's shots have all been flipped
                if (__hip_atomic_load(&ctl->n_flipped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= count) break;
                if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                if (++idle > IDLE_CAP) {
                    if (lane == 0) __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
                continue;
            }
            idle = 0;
        }
        // (diagnostics: groups with a shot to work on, counted after the hand-off above -- a group whose shot has just
        // ended is live again if a prepared one was there for it)
        unsigned long long p_now = 0;
        if (PROF) {
            p_now = (unsigned long long)__builtin_popcountll(__ballot(fs2 != 0 && slot != NONE && gl == 0));
            p_iter++;
            p_live += p_now;
        }
        if (V2) {
            // passes until some group runs out of flips (a shot ends within step_cap passes, so a wave that runs more
            // than that without a shot ending is stuck -- cannot happen; never spin on a bug)
            int32_t left = (int32_t)step_cap;
            if (SV == 4) {
                flip3_passes_asm((const char *)nbt.rec, g.n, wlast, Sg_addr, Fg_addr, Fg_addr + gl * 8u, log_base, log_off, gl, fs2, nfl, left);
            } else {
                do flip3_step_v2((const char *)nbt.rec, g.n, wlast, Sg_addr, Fg_addr, Fg_addr + gl * 8u, log_base, log_off, gl, fs2, nfl);
                while (!__ballot(fs2 == 0) && --left > 0);
                if (left <= 0) left = -1;
            }
            if (PROF) {   // (the loop's top has counted one pass and its live groups)
                const uint32_t more = step_cap - (uint32_t)left - 1u;
                p_live += (unsigned long long)more * p_now;
                p_iter += more;
            }
            if (left < 0) {
                if (lane == 0) __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                break;
            }
            continue;
        }
        flip3_step<PROF, SV>((const char *)nbt.rec, g.n, Sg, Fg, log_base, log_off, gl, fs2, nfl, p_stamp);
        if (__ballot(++steps > step_cap)) {   // cannot happen; never spin on a bug
            if (lane == 0) __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
    }
    if (PROF && lane == 0 && a.counters) {   // flip waves: loop passes, live groups summed over passes, ticks
        atomicAdd(&a.counters[CNT_CAPPED], p_iter | p_live << 32);
        atomicAdd(&a.counters[CNT_XFAIL], __builtin_amdgcn_s_memtime() - p_start);
        // stretches of a pass, 16 bits each, in units of 2^-4 of the wave's total stamped ticks ... kept simple: the
        // shares in 1/1000 packed into the low half of the detected-failure counter's spare neighbour
        const unsigned long long tot = p_stamp[0] + p_stamp[1] + p_stamp[2] + p_stamp[3];
        if (tot && blockIdx.x == 0 && wave == 0)
            printf("[flip4 prof] pass stretches: window read %.3f, search %.3f, row load %.3f, evaluation %.3f of %.1f ticks per pass\n",
                   (double)p_stamp[0] / tot, (double)p_stamp[1] / tot, (double)p_stamp[2] / tot, (double)p_stamp[3] / tot, (double)tot / p_iter);
    }
}


// ---------------------------------------------------------------------------
// Belief propagation.  One workgroup of NW waves owns a group of 64 shot slots;
// in every wave lane = slot = shot.  Messages are laid out [edge][slot] (f64) in
// check-major edge order, so every message access of a wave is one coalesced
// 512-byte row and every graph index is wave-uniform.  The waves of the group
// split the checks (update_checks), the 64-bit chunks of bits (update_bits) and
// the words of the set-up between them and meet at workgroup barriers; the
// per-slot control state (active, iteration, shot id) is replicated in every
// wave and evolves identically.  A slot whose shot stops (syndrome matched or
// iteration cap) is harvested and refilled from a device-wide shot queue.
// ---------------------------------------------------------------------------
constexpr uint32_t REG_DEG = 8;   // degrees up to this keep a row in registers
// (bp_lds_kernel's use of the scalar-constant forms of bp_math.hpp)
#ifndef BPL_TPACK_AHEAD
#define BPL_TPACK_AHEAD 2                // bp_lds_kernel's bit update: a bit's packed graph words loaded one chunk ahead: 1 = bits of one fixed degree
                                        // (per 2^22 shots C2 1921 -> 1813 ms, C1 95.1 -> 89.0 ms), 2 = also mixed degrees up to four (CSS example 122.4 -> 117.5 ms)
#endif
#ifndef BPL_WORDS_BY_BLOCK
#define BPL_WORDS_BY_BLOCK 1            // bp_lds_kernel's BSC words through bsc_block (1) or bsc_word (0): C1 96.1 -> 95.3 ms per 2^22 shots
#endif
#ifndef BPL_LISTED_ERRORS
#define BPL_LISTED_ERRORS 2             // bp_lds_kernel's syndrome phase from a list of the error's set bits: 0 = off, 1 = hybrid form only, 2 = every form with a
                                        // table iteration (per 2^22 shots, one box: C1 101.9 -> 96.4 ms, CSS example 125.0 -> 122.4 ms, C2 within the noise)
#endif
#ifndef BPL_TANH_BATCH
#define BPL_TANH_BATCH 4                // tanh of a bit's three or four messages: 1 = one by one, 2 = in pairs, else as one batch
#endif                                  // (CSS example / C1 per 2^22 shots, one box: 129 / 105.7, 137 / 102.7, 127 / 101.9 ms)


// update_checks for one check of degree DC (belief_propagation.rs:150-166): t = tanh(x/2) of the incoming bit
// messages in ascending bit order (computed by update_bits, which stores bit messages in that form), out = the
// outgoing check messages, flip = 0x80000000 when the check's syndrome bit is set (the reference multiplies by
// -1.0, :160-162).  The edges are worked in batches of HB: enough independent chains to keep the pipeline busy
// next to the other waves of the SIMD, few enough to stay inside 128 registers.
// (HBSEL: batch size; 0 = the sizes measured best for bp_kernel's 128 registers)
template <int DC, int HBSEL = 0, bool SK = false> __device__ inline void bp_check_row(const double (&t)[DC], uint32_t flip, double (&out)[DC])
{
    // (measured: degree 6 in two batches of 3 -- one batch of 6 is no faster, 2 + 2 + 2 is 1 % slower; degree 7 as
    // 4 + 3 -- 6 % faster on the CSS example than 3 + 3 + 1, 2 % faster than one batch of 7)
    constexpr int HB = HBSEL > 0 ? (HBSEL < DC ? HBSEL : DC) : DC <= 3 ? DC : DC == 4 ? 2 : DC == 7 ? 4 : 3;
    double prod = 1.0;
#pragma unroll
    for (int j = 0; j < DC; j++) prod *= t[j];
    // 2 atanh(prod / t_j): the common-case arithmetic of a batch first, the rare operands patched in blocks a
    // wave seldom enters.  |prod| <= |t_j| <= 1, so prod inside the normal band puts every t_j inside it (a zero
    // t_j makes prod zero; a NaN goes through the fast division as through the IEEE one): one test per check.
    const bool odd_prod = bp_exponent_below_band(prod);
#pragma unroll
    for (int h0 = 0; h0 < DC; h0 += HB) {
        double y[HB], w[HB], l[HB];
#pragma unroll
        for (int j = 0; j < HB; j++)
            if (h0 + j < DC) y[j] = bp_div_core(prod, t[h0 + j]);
        if (__builtin_expect(__ballot(odd_prod) != 0, 0)) {
#pragma unroll
            for (int j = 0; j < HB; j++)
                if (h0 + j < DC) y[j] = prod / t[h0 + j];
        }
        bool rare = false, special = false;
#pragma unroll
        for (int j = 0; j < HB; j++)
            if (h0 + j < DC) {
                w[j] = bp_div_core(2.0 * y[j], 1.0 - y[j]);   // bp_atanh_arg without its select
                rare |= y[j] == 1.0;
            }
        if (__builtin_expect(__ballot(rare) != 0, 0)) {        // 2 / 0 = +inf (bp_math.hpp, bp_atanh_arg)
#pragma unroll
            for (int j = 0; j < HB; j++)
                if (h0 + j < DC) w[j] = y[j] == 1.0 ? __builtin_huge_val() : w[j];
        }
#pragma unroll
        for (int j = 0; j < HB; j++)
            if (h0 + j < DC) {
                special |= bp_log1p_is_special(w[j]);
                l[j] = bp_log1p_main<SK>(w[j]);
            }
        if (__builtin_expect(__ballot(special) != 0, 0)) {
#pragma unroll
            for (int j = 0; j < HB; j++)
                if (h0 + j < DC) l[j] = bp_log1p_fix(w[j], l[j]);
        }
#pragma unroll
        for (int j = 0; j < HB; j++)
            if (h0 + j < DC) {
                const double v = 2.0 * (0.5 * l[j]);
                out[h0 + j] = __hiloint2double(__double2hiint(v) ^ (int)flip, __double2loint(v));
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// update_bits (belief_propagation.rs:168-180) + hard decision (:100-113) for the 64 bits of one chunk, all 64
// slots of the group (lane = slot).  A bit message m = (sum - check msg) + prior is stored as tanh(m/2), the only
// form update_checks uses it in (:154,158): the f64 work of an iteration is split between the check update
// (division, atanh) and this HBM-bound phase.  DV > 0: every bit has exactly DV checks.
template <int DV>
__device__ inline void bp_bits_chunk(const DeviceGraph &g, uint32_t item, uint32_t split, uint32_t lane, bool active,
                                     uint64_t actmask, double prior, const double *chkm, double *bitsm, uint64_t *dw,
                                     uint64_t *dsl, uint64_t *psl = nullptr)
{
    // item = chunk * split + piece: a 64-bit chunk is worked in `split` (1, 2 or 4) pieces, so that codes with
    // few chunks still spread evenly over the waves of a group
    const uint32_t n = g.n;
    uint64_t dacc = 0, slice = 0;
    const uint32_t chunk = item / split, piece = item % split, width = 64 / split;
    const uint32_t b0 = chunk * 64 + piece * width;
    if (b0 >= n) return;
    const uint32_t b1 = b0 + width < n ? b0 + width : n;
    if constexpr (DV > 0) {
        // the chunk's edge lists are fetched once, lane i holding those of bit b0 + i; the loop below
        // reads them back as scalars (readlane), so every row address is scalar base + lane offset and
        // no row load waits for an index load.  Software pipeline over blocks of BB bits: the BB*DV
        // rows of the next block are in flight while the current block is summed and written back.
        constexpr uint32_t BB = 4;
        uint32_t myed[DV];
#pragma unroll
        for (int k = 0; k < DV; k++) myed[k] = b0 + lane < b1 ? g.t_edge[(b0 + lane) * (uint32_t)DV + k] : 0u;
        const uint32_t nbits = b1 - b0;
        double vn[BB][DV];
#pragma unroll
        for (uint32_t i = 0; i < BB; i++)
            if (i < nbits) {
#pragma unroll
                for (int k = 0; k < DV; k++)
                    vn[i][k] = chkm[(size_t)__builtin_amdgcn_readlane(myed[k], i) * 64 + lane];
            }
        for (uint32_t i0 = 0; i0 < nbits; i0 += BB) {
            double v[BB][DV];
#pragma unroll
            for (uint32_t i = 0; i < BB; i++)
#pragma unroll
                for (int k = 0; k < DV; k++) v[i][k] = vn[i][k];
#pragma unroll
            for (uint32_t i = 0; i < BB; i++)
                if (i0 + BB + i < nbits) {
#pragma unroll
                    for (int k = 0; k < DV; k++)
                        vn[i][k] = chkm[(size_t)__builtin_amdgcn_readlane(myed[k], i0 + BB + i) * 64 + lane];
                }
#pragma unroll
            for (uint32_t i = 0; i < BB; i++)
                if (i0 + i < nbits) {
                    double sum = 0.0, post = prior;
#pragma unroll
                    for (int k = 0; k < DV; k++) {
                        sum += v[i][k];
                        post += v[i][k];
                    }
                    double tm[DV];
#pragma unroll
                    for (int k = 0; k < DV; k++) tm[k] = bp_tanh_half(sum - v[i][k] + prior);
#pragma unroll
                    for (int k = 0; k < DV; k++)
                        if (active)
                            bitsm[(size_t)__builtin_amdgcn_readlane(myed[k], i0 + i) * 64 + lane] = tm[k];
                    const bool dec = post < 0.0;
                    dacc |= (uint64_t)dec << (i0 + i);
                    const uint64_t mask = __ballot(active && dec);
                    if (lane == i0 + i) slice = mask;
                }
        }
    } else {
        // any degrees (DV = 0), or any degrees up to -DV (DV < 0: the row registers are sized for that, half of the
        // general eight when the code has no bit of degree above 4).  The chunk's edge-list bounds (lane i = bit b0 + i) and, when they fit four registers
        // (<= 256 edges in the chunk), the edge list itself are fetched once and read back as scalars, so row
        // addresses are scalar and no row load waits on an index load; the rows of the next bit are in flight
        // while the current bit is worked.
        constexpr uint32_t RD = DV < 0 ? (uint32_t)(-DV) : REG_DEG;
        const uint32_t nbits = b1 - b0;
        const uint32_t myk0 = b0 + lane < b1 ? g.t_ptr[b0 + lane] : 0u, myk1 = b0 + lane < b1 ? g.t_ptr[b0 + lane + 1] : 0u;
        const uint32_t kstart = __builtin_amdgcn_readlane(myk0, 0), cnt = __builtin_amdgcn_readlane(myk1, nbits - 1) - kstart;
        const bool listed = cnt <= 256;
        uint32_t me[4] = {0, 0, 0, 0};
        if (listed) {
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)
                if (lane + 64 * j < cnt) me[j] = g.t_edge[kstart + lane + 64 * j];
        }
        auto edge = [&](uint32_t k) -> uint32_t {   // k wave-uniform
            if (!listed) return __builtin_amdgcn_readfirstlane(g.t_edge[k]);
            const uint32_t o = k - kstart, l = o & 63, qd = o >> 6;
            const uint32_t a0 = __builtin_amdgcn_readlane(me[0], l), a1 = __builtin_amdgcn_readlane(me[1], l),
                           a2 = __builtin_amdgcn_readlane(me[2], l), a3 = __builtin_amdgcn_readlane(me[3], l);
            return qd == 0 ? a0 : qd == 1 ? a1 : qd == 2 ? a2 : a3;
        };
        double vn[RD];
        uint32_t edn[RD];
#pragma unroll
        for (uint32_t k = 0; k < RD; k++) {
            vn[k] = 0.0;
            edn[k] = 0;
        }
        uint32_t k0n = kstart, dvn = nbits ? __builtin_amdgcn_readlane(myk1, 0) - kstart : 0u;
        if (dvn <= RD) {
#pragma unroll
            for (uint32_t k = 0; k < RD; k++)
                if (k < dvn) {
                    edn[k] = edge(k0n + k);
                    vn[k] = chkm[(size_t)edn[k] * 64 + lane];
                }
        }
        for (uint32_t i = 0; i < nbits; i++) {
            const uint32_t k0 = k0n, dv = dvn;
            double sum = 0.0, post = prior;
            double v[RD];
            uint32_t ed[RD];
#pragma unroll
            for (uint32_t k = 0; k < RD; k++) {
                v[k] = vn[k];
                ed[k] = edn[k];
            }
            if (i + 1 < nbits) {
                k0n = __builtin_amdgcn_readlane(myk0, i + 1);
                dvn = __builtin_amdgcn_readlane(myk1, i + 1) - k0n;
                if (dvn <= RD) {
#pragma unroll
                    for (uint32_t k = 0; k < RD; k++)
                        if (k < dvn) {
                            edn[k] = edge(k0n + k);
                            vn[k] = chkm[(size_t)edn[k] * 64 + lane];
                        }
                }
            }
            if (dv <= RD) {
#pragma unroll
                for (uint32_t k = 0; k < RD; k++)
                    if (k < dv) {
                        sum += v[k];
                        post += v[k];
                    }
                double tm[RD];
#pragma unroll
                for (uint32_t k = 0; k < RD; k++)
                    if (k < dv) tm[k] = bp_tanh_half(sum - v[k] + prior);
#pragma unroll
                for (uint32_t k = 0; k < RD; k++)
                    if (k < dv && active) bitsm[(size_t)ed[k] * 64 + lane] = tm[k];
            } else {
                for (uint32_t k = 0; k < dv; k++) {
                    const double vv = chkm[(size_t)edge(k0 + k) * 64 + lane];
                    sum += vv;
                    post += vv;
                }
                for (uint32_t k = 0; k < dv; k++) {
                    const size_t off = (size_t)edge(k0 + k) * 64 + lane;
                    const double tm = bp_tanh_half(sum - chkm[off] + prior);
                    if (active) bitsm[off] = tm;
                }
            }
            const bool dec = post < 0.0;
            dacc |= (uint64_t)dec << i;
            const uint64_t mask = __ballot(active && dec);
            if (lane == i) slice = mask;
        }
    }
    if (active) {
        if (split == 1) dw[(size_t)chunk * 64 + lane] = dacc;
        else if (split == 2) ((uint32_t *)dw)[((size_t)chunk * 64 + lane) * 2 + piece] = (uint32_t)dacc;
        else ((uint16_t *)dw)[((size_t)chunk * 64 + lane) * 4 + piece] = (uint16_t)dacc;
    }
    const uint32_t bb = b0 + lane;
    if (bb < b1) {
        const uint64_t old = dsl[bb];
        dsl[bb] = (old & ~actmask) | slice;
        // streamed form: the stop test's residual syndrome ^ H * decision follows the decisions that changed (few after
        // the first iterations), instead of being gathered anew from every bit's row in every iteration
        const uint64_t chg = (old & actmask) ^ slice;
        if (psl && chg)
            for (uint32_t k = g.t_ptr[bb]; k < g.t_ptr[bb + 1]; k++) atomicXor((unsigned long long *)&psl[g.t_idx[k]], (unsigned long long)chg);
    }
}

// DC / DV > 0: every check has exactly DC bits / every bit exactly DV checks (regular code):
// rows start at c*DC, loops are fully unrolled and the DC (DV) independent tanh / log1p chains of
// a row interleave.  0 = any degrees (row pointers, runtime trip counts).
template <int DC, int DV> __global__ __launch_bounds__(512, 4) void bp_kernel(DeviceGraph g, BpArgs a)
{
    __shared__ uint32_t q[65];
    __shared__ unsigned long long sh_queue, sh_mism;
    __shared__ uint32_t sh_flag[2][64];   // per slot: [0] residual non-zero, [1] a logical generator anticommutes
    // wv is wave-uniform: tell the compiler, so that check / bit / edge indices live in SGPRs and the
    // graph arrays are read with scalar loads
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), NW = blockDim.x >> 6, NT = blockDim.x;
    for (uint32_t i = tid; i < 65; i += NT) q[i] = a.gap.q[i];
    __syncthreads();
    const uint32_t n = g.n, m = g.m;
    const uint32_t Wn = (n + 63) / 64, Wm = (m + 63) / 64;
    const size_t grp = blockIdx.x;
    double *bitsm = a.msg + grp * 2 * (size_t)g.E * 64;
    double *chkm = bitsm + (size_t)g.E * 64;
    uint64_t *ew = a.ew + grp * (size_t)Wn * 64;
    uint64_t *dw = a.dw + grp * (size_t)Wn * 64;
    uint64_t *dsl = a.dsl + grp * (size_t)n;
    uint64_t *ssl = a.ssl + grp * (size_t)m;
    uint64_t *esl = a.esl + grp * (size_t)n;
    const double prior = a.prior;
    const double t_prior = bp_tanh_half(prior);   // the bit messages of a fresh slot are the prior (:62-66)

    // replicated per-slot state (identical in all waves of the group)
    bool drained = false, active = false, fresh = false;
    uint32_t iter = 0;
    uint64_t myi = 0;
    uint64_t c_shots = 0, c_fail = 0, c_det = 0, c_work = 0;   // accumulated by wave 0 only

    // LDPC_HIP_BP_PROFILE: cycles of block 0 / wave 0 per phase (refill, stop test, harvest, checks, bits)
    uint64_t prof[5] = {0, 0, 0, 0, 0}, tprev = a.profile ? __builtin_readcyclecounter() : 0;
    auto lap = [&](int which) {
        if (a.profile) { const uint64_t tnow = __builtin_readcyclecounter(); prof[which] += tnow - tprev; tprev = tnow; }
    };
    for (;;) {
        // ---- refill idle slots from the device-wide queue (a.queue = shots handed out so far) ----
        {
            const uint64_t idle = __ballot(!active);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0));
            uint64_t first_taken = a.n_shots;
            if (idle && !drained) {
                if (tid == 0) sh_queue = atomicAdd(a.queue, (unsigned long long)__builtin_popcountll(idle));
                __syncthreads();
                first_taken = sh_queue;
                __syncthreads();
                if (first_taken >= a.n_shots) { first_taken = a.n_shots; drained = true; }
            }
            const bool take = !active && first_taken + rank < a.n_shots;
            const uint64_t takemask = __ballot(take);
            if (takemask) {
                if (take) { myi = first_taken + rank; active = true; fresh = true; iter = 0; }
                if (first_taken + (uint64_t)__builtin_popcountll(idle) >= a.n_shots) drained = true;
                const uint64_t shot = a.first_shot + myi;
                if (a.source != SRC_SYNDROMES) {
                    // error words of the taken slots, kept per slot (ew, for the failure test) and transposed
                    // to bit-sliced rows esl[b] = mask of the taken slots whose word has bit b (64 x 64 bit
                    // transpose by ballot; only bit positions some slot has set are visited)
                    for (uint32_t w = wv; w < Wn; w += NW) {
                        uint64_t x = 0;
                        if (take) {
                            uint32_t nb = bits_in_word(n, w);
                            if (a.source == SRC_BSC) x = bsc_word(q, a.gap.inv_log2q, a.gap.p_is_zero, a.seed, shot, w, nb);
                            else if (a.source == SRC_WORDS) {
                                x = a.words_in[myi * Wn + w];
                                if (nb < 64) x &= (1ull << nb) - 1;
                            } else {
                                uint64_t xx, zz;
                                depol_word(q, a.gap.inv_log2q, a.gap.p_is_zero, a.seed, shot, w, nb, xx, zz);
                                x = a.source == SRC_DEPOL_X ? xx : zz;
                            }
                            ew[(size_t)w * 64 + lane] = x;
                        }
                        const uint64_t anyv = wave_or64(x);
                        uint64_t any = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(anyv >> 32)) << 32) |
                                       __builtin_amdgcn_readfirstlane((uint32_t)anyv);
                        uint64_t row = 0;
                        while (any) {
                            const uint32_t i = (uint32_t)__builtin_ctzll(any);
                            any &= any - 1;
                            const uint64_t mask = __ballot((x >> i) & 1ull);
                            if (lane == i) row = mask;
                        }
                        const uint32_t b = w * 64 + lane;
                        if (b < n) esl[b] = row;
                    }
                    __syncthreads();
                    // syndromes of the taken slots, all at once: XOR of the bit-sliced rows of a check's bits
                    for (uint32_t c = tid; c < m; c += NT) {
                        uint64_t par = 0;
                        if constexpr (DC > 0) {
                            uint32_t bi[DC];
#pragma unroll
                            for (int j = 0; j < DC; j++) bi[j] = g.h_idx[c * (uint32_t)DC + j];
#pragma unroll
                            for (int j = 0; j < DC; j++) par ^= esl[bi[j]];
                        } else {
                            for (uint32_t e = g.h_ptr[c]; e < g.h_ptr[c + 1]; e++) par ^= esl[g.h_idx[e]];
                        }
                        ssl[c] = (ssl[c] & ~takemask) | (par & takemask);
                    }
                } else {
                    for (uint32_t c = wv; c < m; c += NW) {
                        uint32_t sb = 0;
                        if (take) sb = (uint32_t)(a.synd_in[myi * Wm + (c >> 6)] >> (c & 63)) & 1u;
                        uint64_t mask = __ballot(take && sb);
                        if (lane == 0) ssl[c] = (ssl[c] & ~takemask) | mask;
                    }
                }
                // decision before the first update: posterior = prior (belief_propagation.rs:100-113)
                const bool init = prior < 0.0;
                for (uint32_t b = tid; b < n; b += NT) dsl[b] = (dsl[b] & ~takemask) | (init ? takemask : 0ull);
                if (take)
                    for (uint32_t w = wv; w < Wn; w += NW) {
                        uint32_t nb = bits_in_word(n, w);
                        dw[(size_t)w * 64 + lane] = init ? (nb < 64 ? (1ull << nb) - 1 : ~0ull) : 0ull;
                    }
                __syncthreads();
            }
        }
        const uint64_t actmask = __ballot(active);
        if (!actmask) break;
        lap(0);

        // (The streamed form keeps syndrome ^ H * decision up to date from the bit update instead, bps_stop; tried here
        // too: at C2 the decision rows are L2-resident, the gathers cheap, and the extra atomics cost 5 %.)
        // ---- stop test (belief_propagation.rs:26-27), bit-sliced over the 64 slots: the threads of
        // the group split the checks, each XORs whole 64-slot words; partial masks are OR-ed in LDS
        if (tid == 0) sh_mism = 0;
        __syncthreads();
        {
            uint64_t part = 0;
            for (uint32_t c = tid; c < m; c += NT) {
                uint64_t x = ssl[c];
                if constexpr (DC > 0) {
                    // regular rows: the DC indices, then the DC rows, are fetched together (two memory
                    // latencies per check instead of 2 DC)
                    uint32_t bi[DC];
#pragma unroll
                    for (int j = 0; j < DC; j++) bi[j] = g.h_idx[c * (uint32_t)DC + j];
#pragma unroll
                    for (int j = 0; j < DC; j++) x ^= dsl[bi[j]];
                } else {
                    for (uint32_t e = g.h_ptr[c]; e < g.h_ptr[c + 1]; e++) x ^= dsl[g.h_idx[e]];
                }
                part |= x;
            }
            part = wave_or64(part);
            if (lane == 0 && part) atomicOr(&sh_mism, (unsigned long long)part);
        }
        __syncthreads();
        const uint64_t mism = sh_mism;
        const bool conv = !((mism >> lane) & 1ull);
        const bool done = active && (conv || iter >= a.max_iter);
        const uint64_t donemask = __ballot(done);
        lap(1);

        if (donemask) {
            // ---- harvest ------------------------------------------------------
            bool fail = false, det = !conv;
            if (a.test != TEST_NONE) {
                for (uint32_t i = tid; i < 128; i += NT) sh_flag[i >> 6][i & 63] = 0;
                __syncthreads();
                uint64_t rany = 0;
                for (uint32_t w = wv; w < Wn; w += NW)
                    if (done) rany |= ew[(size_t)w * 64 + lane] ^ dw[(size_t)w * 64 + lane];
                if (rany) atomicOr(&sh_flag[0][lane], 1u);
                if (a.test == TEST_LOGICAL) {
                    uint32_t lfail = 0;
                    for (uint32_t r = wv; r < a.l_rows; r += NW) {
                        uint32_t par = 0;
                        for (uint32_t k = a.l_ptr[r]; k < a.l_ptr[r + 1]; k++) {
                            uint32_t b = a.l_idx[k];
                            if (done) {
                                size_t off = (size_t)(b >> 6) * 64 + lane;
                                par ^= (uint32_t)((ew[off] ^ dw[off]) >> (b & 63)) & 1u;
                            }
                        }
                        lfail |= par;
                    }
                    if (lfail) atomicOr(&sh_flag[1][lane], 1u);
                }
                __syncthreads();
                if (a.test == TEST_RESIDUAL) fail = sh_flag[0][lane] != 0;
                else fail = det || sh_flag[1][lane] != 0;
            }
            if (done) {
                if (a.corr_out)
                    for (uint32_t w = wv; w < Wn; w += NW) {
                        uint64_t x = dw[(size_t)w * 64 + lane];
                        if (a.xor_input) x ^= ew[(size_t)w * 64 + lane];
                        a.corr_out[myi * Wn + w] = x;
                    }
                if (a.post_out)
                    for (uint32_t b = wv; b < n; b += NW) {      // BpState::decode posterior, :100-106
                        double l = prior;
                        if (!fresh)
                            for (uint32_t k = g.t_ptr[b]; k < g.t_ptr[b + 1]; k++) l += chkm[(size_t)g.t_edge[k] * 64 + lane];
                        a.post_out[myi * n + b] = l;
                    }
                if (a.chk_out)   // check -> bit messages at stop (belief_propagation.rs:156-163), Edges order; 0.0 before any update (:68-76)
                    for (uint32_t e = wv; e < g.E; e += NW) a.chk_out[myi * g.E + e] = fresh ? 0.0 : chkm[(size_t)e * 64 + lane];
                if (wv == 0) {
                    if (a.iters_out) a.iters_out[myi] = iter;
                    if (a.failed_out && fail) a.failed_out[myi] |= (uint8_t)a.fail_bit;
                    c_shots++; c_fail += fail; c_det += det; c_work += iter;
                }
                active = false;
            }
            __syncthreads();
            lap(2);
            if (!drained) continue;     // refill before spending an iteration
            if (!__ballot(active)) break;
        }
        // tail hand-off: nothing left to refill with and only long-running shots remain in this group
        if (drained && a.handoff_iter != 0xFFFFFFFFu && __ballot(active) && !__ballot(active && iter < a.handoff_iter)) {
            if (wv == 0) {
                const uint64_t am = __ballot(active), fm = __ballot(active && fresh);
                a.handoff.iter[grp * 64 + lane] = iter;
                a.handoff.myi[grp * 64 + lane] = myi;
                if (lane == 0) {
                    a.handoff.act[grp] = am;
                    a.handoff.fresh[grp] = fm;
                    atomicExch(a.handoff_flag, 1ull);
                }
            }
            break;
        }

        // ---- one BP iteration on the active slots (update_once, :133-140) -----
        // update_checks (:150-166): the waves take the checks round-robin
        if constexpr (DC > 0) {
            // software pipeline: the DC message rows of the wave's next check are in flight while the
            // tanh / log1p chains of the current one run
            // (DC >= 7: the second row of registers does not fit next to the batch temporaries; the row is loaded
            // where it is used and the other waves of the SIMD cover the latency)
            constexpr bool PF = DC < 7;
            double xn[PF ? DC : 1];
            if (PF && wv < m) {
#pragma unroll
                for (int j = 0; j < (PF ? DC : 0); j++) xn[j] = fresh ? t_prior : bitsm[((size_t)wv * DC + j) * 64 + lane];
            }
            for (uint32_t c = wv; c < m; c += NW) {
                const size_t e0 = (size_t)c * DC;
                const bool sbit = (ssl[c] >> lane) & 1ull;
                double x[DC];
                if constexpr (PF) {
#pragma unroll
                    for (int j = 0; j < DC; j++) x[j] = xn[j];
                    if (c + NW < m) {
#pragma unroll
                        for (int j = 0; j < DC; j++) xn[j] = fresh ? t_prior : bitsm[((size_t)(c + NW) * DC + j) * 64 + lane];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < DC; j++) x[j] = fresh ? t_prior : bitsm[(e0 + j) * 64 + lane];
                }
                double out[DC];
                bp_check_row<DC>(x, sbit ? 0x80000000u : 0u, out);
#pragma unroll
                for (int j = 0; j < DC; j++)
                    if (active) chkm[(e0 + j) * 64 + lane] = out[j];
            }
        } else {
            for (uint32_t c = wv; c < m; c += NW) {
                const uint32_t e0 = g.h_ptr[c], dc = g.h_ptr[c + 1] - e0;
                const bool sbit = (ssl[c] >> lane) & 1ull;
                if (dc <= REG_DEG) {
                    double t[REG_DEG];
                    double prod = 1.0;
#pragma unroll
                    for (uint32_t j = 0; j < REG_DEG; j++)
                        if (j < dc) {
                            t[j] = fresh ? t_prior : bitsm[(size_t)(e0 + j) * 64 + lane];
                            prod *= t[j];
                        }
#pragma unroll
                    for (uint32_t j = 0; j < REG_DEG; j++)
                        if (j < dc) {
                            double v = 2.0 * bp_ref_atanh(bp_div_prod(prod, t[j]));
                            if (sbit) v *= -1.0;
                            if (active) chkm[(size_t)(e0 + j) * 64 + lane] = v;
                        }
                } else {
                    double prod = 1.0;
                    for (uint32_t j = 0; j < dc; j++) prod *= fresh ? t_prior : bitsm[(size_t)(e0 + j) * 64 + lane];
                    for (uint32_t j = 0; j < dc; j++) {
                        const double t = fresh ? t_prior : bitsm[(size_t)(e0 + j) * 64 + lane];
                        double v = 2.0 * bp_ref_atanh(bp_div_prod(prod, t));
                        if (sbit) v *= -1.0;
                        if (active) chkm[(size_t)(e0 + j) * 64 + lane] = v;
                    }
                }
            }
        }
        __syncthreads();
        lap(3);
        // update_bits (:168-180) + hard decision (:100-113): the waves take 64-bit chunks round-robin
        {
            const uint32_t split = Wn >= 4 * NW ? 1u : Wn >= 2 * NW ? 2u : 4u;
            for (uint32_t item = wv; item < Wn * split; item += NW)
                bp_bits_chunk<DV>(g, item, split, lane, active, actmask, prior, chkm, bitsm, dw, dsl);
        }
        if (active) { iter++; fresh = false; }
        __syncthreads();
        lap(4);
    }
    if (a.profile && blockIdx.x == 0 && tid == 0)
        printf("bp profile (block 0, wave 0) cycles: refill %llu stop-test %llu harvest %llu checks %llu bits %llu\n", (unsigned long long)prof[0],
               (unsigned long long)prof[1], (unsigned long long)prof[2], (unsigned long long)prof[3], (unsigned long long)prof[4]);
    if (wv == 0) {
        c_shots = wave_sum64(c_shots); c_fail = wave_sum64(c_fail); c_det = wave_sum64(c_det); c_work = wave_sum64(c_work);
        if (lane == 0 && a.counters) {
            if (c_shots) atomicAdd(&a.counters[CNT_SHOTS], (unsigned long long)c_shots);
            if (c_fail) atomicAdd(&a.counters[CNT_FAIL], (unsigned long long)c_fail);
            if (c_det) {
                atomicAdd(&a.counters[CNT_DET], (unsigned long long)c_det);
                atomicAdd(&a.counters[CNT_CAPPED], (unsigned long long)c_det);
            }
            if (c_fail - c_det && a.test == TEST_RESIDUAL) atomicAdd(&a.counters[CNT_UNDET], (unsigned long long)(c_fail - c_det));
            if (c_work) atomicAdd(&a.counters[CNT_WORK], (unsigned long long)c_work);
        }
    }
}


// ---------------------------------------------------------------------------
// Belief propagation with the messages of a shot in LDS (small codes: E * 8 bytes beside the shot's words fit a
// fraction of a CU's 160 KiB).  One workgroup per shot in flight, shots from the device-wide queue; a thread works a
// check in update_checks and a bit in update_bits.  ONE message array serves both directions: a check's DC entries
// belong to that check alone and a bit's entries to that bit alone, so update_checks overwrites the tanh of the
// incoming bit messages with the outgoing check messages in place, and update_bits overwrites those with the tanh of
// the new bit messages.  The arithmetic is bp_kernel's (bp_check_row, bp_tanh_half, the same orders of the products
// and sums), so the results are bit-identical to it; nothing of an iteration touches HBM, and a shot that does not
// converge holds one workgroup, not a 64-slot group.  Posterior / message outputs are left to the other forms.
// ---------------------------------------------------------------------------
// Four waves per shot.  (Five, 320 threads, leave fewer lanes idle on 300 / 900 checks and 625 / 1200 bits, C3 / C1, and
// were the form of most of round 2; with four, six or seven shots fit a CU instead of five, and the phases of a shot
// that run on few lanes -- sampling, stop tests, the failure test -- hide behind more neighbours: CSS example 131 ->
// 125 ms, C1 102.5 -> 101.8 ms per 2^22 shots on one box.  Three waves: 125 / 118; two: 131 / 144.)
#ifndef BPL_NT
#define BPL_NT 256
#endif
constexpr uint32_t BPL_THREADS = BPL_NT, BPL_MAX_THREADS = 1024;
// first-iteration table (see the kernel): bit messages [degree 1..4][sign pattern][edge] and posteriors [degree 0..4][pattern]
constexpr uint32_t BPL_TABLE_WORDS = 4 * 16 * 4 + 5 * 16;
#ifndef BPL_CHECK_BATCH
#define BPL_CHECK_BATCH 0                // bp_kernel's batch sizes (batches of 2 spill less at 72 registers and measured slower: CSS 161 -> 216 ms)
#endif
#ifndef BPL_PERIOD_FROM
#define BPL_PERIOD_FROM 6               // first (even) iteration whose bit messages are kept for the period test (CSS example:
                                        // 2 and 4 end no run earlier and cost saves, 138 / 129 against 127.6 ms; 8 ends runs later, 129)
#endif
#ifndef BPL_SK
#define BPL_SK true                     // polynomial constants built at their use by scalar moves (bp_math.hpp)
#endif
#ifndef BPL_HYB_WAVES_PER_SIMD
#define BPL_HYB_WAVES_PER_SIMD 7        // (the hybrid form runs one workgroup of 16 waves per CU and could have 128 registers per lane:
                                        // measured 531 against 524.5 ms per 2^20 shots of C2 with 72)
#endif
#ifndef BPL_WAVES_PER_SIMD
#define BPL_WAVES_PER_SIMD 7            // register budget: 72 per lane, so that seven four-wave workgroups fit a CU when their LDS does
                                        // (with five-wave workgroups, CSS example / C1 ms per 2^22 shots: 4 waves 225 / 102 and 5 waves 167 / 103 with the
                                        // compiler's constants, 5 waves 164 / 101, 7 waves 131 / 102, 8 waves, six workgroups, 148 / 103)
#endif

// LDS of a workgroup, in u64 words: messages, error / decision / syndrome / residual words, gap table, queue slot,
// then copies of the graph as u16 when they do not cost a resident workgroup (copies >= 1: bit -> edges, and bit ->
// first edge for codes with bits of mixed degree; 2: bit -> checks as well; 3: the logical generators' rows of the CSS test)
__host__ __device__ inline size_t bp_lds_words(const DeviceGraph &g, uint32_t copies, bool mixed_bits, uint32_t l_rows, uint32_t l_nnz,
                                               uint32_t e_lds = 0xFFFFFFFFu)   // (edges whose messages live in LDS: all, but for the hybrid form)
{
    const size_t Wn = (g.n + 63) / 64, Wm = (g.m + 63) / 64;
    // (copies == 4: the logical generators' rows alone -- the kernels that load a bit's packed graph words a chunk ahead
    // from global memory have no use for the graph in LDS)
    const bool graph1 = copies >= 1 && copies <= 3, graph2 = copies >= 2 && copies <= 3;
    const size_t u16s = (graph1 ? (size_t)g.E + (mixed_bits ? g.n + 1 : 0) : 0) + (graph2 ? (size_t)g.E : 0) +
                        (copies >= 3 ? (size_t)l_rows + 1 + l_nnz : 0);
    return (size_t)std::min(g.E, e_lds) + 2 * Wn + 2 * Wm + 34 + 2 + 5 + BPL_TABLE_WORDS + (u16s + 3) / 4;
}

// (diagnostic build, -DBPL_PROF: wave 0 of workgroup 0 prints the cycles it spent in each phase)
#ifdef BPL_PROF
#define BPL_STAMP(k) do { const uint64_t tn_ = __builtin_readcyclecounter(); prof[k] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define BPL_STAMP(k) do { } while (0)
#endif

// The lane index, computed where it is asked for.  Everything a phase of bp_lds_kernel derives from its thread index
// (addresses of its words, rows, shuffle partners) is the same for every shot, so the compiler computes it once at the
// top of the kernel, cannot keep it in 72 registers across the check rows, and reads it back from scratch memory in
// every phase of every shot: a few hundred cycles of latency each time, in series.  A phase that starts from this
// (volatile: neither hoisted nor shared) recomputes those values with a handful of instructions instead.
__device__ inline uint32_t bpl_lane()
{
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
#define BPL_FRESH() const uint32_t lane = bpl_lane(), tid = wv * 64u + lane

// DV: every bit has exactly DV checks (> 0), at most -DV (< 0), or any number (0)
template <int DC, int DV, bool WIDE, bool HYB = false>
__global__ __launch_bounds__(WIDE ? BPL_MAX_THREADS : BPL_THREADS, HYB ? BPL_HYB_WAVES_PER_SIMD : BPL_WAVES_PER_SIMD) void bp_lds_kernel(DeviceGraph g, BpArgs a, uint32_t copies, uint32_t e_lds)
{
    extern __shared__ uint64_t lds[];
    const uint32_t tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // (the phases of a shot take their own, BPL_FRESH)
    // (WIDE: a multiple of 64 chosen by bp_lds_shape for codes of which few shots fit a CU; the usual 256 as a constant
    // is worth 2 % on C1)
    const uint32_t NT = WIDE ? blockDim.x : BPL_THREADS;
    const uint32_t n = g.n, m = g.m, E = g.E, Wn = (n + 63) / 64, Wm = (m + 63) / 64;
    // (HYB: a shot's messages do not fit LDS; those of the edges from e_lds on live in this workgroup's piece of a.msg,
    // L2-resident: one workgroup per CU, a few MB per XCD)
    const uint32_t EL = HYB ? e_lds : E;
    double *msg = (double *)lds;
    uint64_t *ew = lds + EL, *dw = ew + Wn, *sw = dw + Wn, *rw = sw + Wm;
    uint32_t *q = (uint32_t *)(rw + Wm);
    uint32_t *sh_flag = q + 65, *sh_fail = q + 66;   // (the words after the gap table's 65)
    unsigned long long *sh_queue = (unsigned long long *)(rw + Wm + 34);   // (two slots, taken in turn: one barrier per hand-over)
    // (thread 0's counters live in LDS, not in ten registers that every check row would have to keep or spill)
    unsigned long long *cnt = sh_queue + 2;   // shots, failures, runs that did not converge, iterations, iterations of arithmetic
    double *tab_t = (double *)(cnt + 5), *tab_p = tab_t + 4 * 16 * 4;
    uint16_t *tedge = (uint16_t *)(tab_p + 5 * 16), *tptr = tedge + E, *tidx = tptr + (DV > 0 ? 0 : n + 1);
    const bool c_edges = copies >= 1 && copies <= 3, c_checks = copies >= 2 && copies <= 3, c_logicals = copies >= 3;   // (4: logical rows alone)
    uint16_t *lptr = copies == 4 ? tedge : tidx + (c_checks ? E : 0), *lidx = lptr + a.l_rows + 1;
    auto edge_of = [&](uint32_t k) -> uint32_t { return c_edges ? (uint32_t)tedge[k] : g.t_pack[k] & 0xFFFFu; };
    auto first_edge = [&](uint32_t b) -> uint32_t { return c_edges ? (uint32_t)tptr[b] : g.t_ptr[b]; };
    auto check_of = [&](uint32_t k) -> uint32_t { return c_checks ? (uint32_t)tidx[k] : g.t_pack[k] >> 16; };
    auto slot_of = [&](uint32_t k) -> uint32_t {   // edge | check << 16
        return c_checks ? (uint32_t)tedge[k] | ((uint32_t)tidx[k] << 16) : c_edges ? (uint32_t)tedge[k] | (g.t_pack[k] & 0xFFFF0000u) : g.t_pack[k];
    };
    uint32_t *dw32 = (uint32_t *)dw, *rw32 = (uint32_t *)rw, *sw32 = (uint32_t *)sw;
    const uint32_t *ew32 = (const uint32_t *)ew;
    for (uint32_t i = tid; i < 65; i += NT) q[i] = a.gap.q[i];
    if (c_edges) {
        for (uint32_t e = tid; e < E; e += NT) tedge[e] = (uint16_t)g.t_pack[e];
        if (DV <= 0)
            for (uint32_t b = tid; b <= n; b += NT) tptr[b] = (uint16_t)g.t_ptr[b];
    }
    if (c_checks)
        for (uint32_t e = tid; e < E; e += NT) tidx[e] = (uint16_t)(g.t_pack[e] >> 16);
    if (c_logicals) {
        for (uint32_t r = tid; r <= a.l_rows; r += NT) lptr[r] = (uint16_t)a.l_ptr[r];
        for (uint32_t k = tid; k < a.l_nnz; k += NT) lidx[k] = (uint16_t)a.l_idx[k];
    }
    const double prior = a.prior, t_prior_v = bp_tanh_half<BPL_SK>(prior);
    // (wave-uniform values are moved to scalar registers by hand where the compiler would hold them per lane)
    const double t_prior = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(t_prior_v)),
                                            __builtin_amdgcn_readfirstlane(__double2loint(t_prior_v)));
    const bool init = prior < 0.0;   // decision before the first update: posterior = prior (belief_propagation.rs:100-113)
    // The first iteration of every shot starts from the same messages (the prior, :60-68), so update_checks gives one
    // value v0 for every edge, negated where the check's syndrome bit is set (:160-162), and what update_bits makes of
    // a bit's d <= 4 incoming +-v0 depends on their signs alone: 2^d cases.  They are computed once per workgroup, by
    // the functions and in the orders of the general path (so bit for bit what it would compute), and the first
    // iteration of a shot is then a table look-up per edge.  (DV == 0, bits of any degree: general path throughout.)
    constexpr bool FIRST_BY_TABLE = DV != 0 && DC > 0;
    if constexpr (FIRST_BY_TABLE) {
        double x[DC], out[DC];
#pragma unroll
        for (int j = 0; j < DC; j++) x[j] = t_prior;
        bp_check_row<DC, BPL_CHECK_BATCH, BPL_SK>(x, 0u, out);
        const double v0 = out[0];
        for (uint32_t t5 = tid; t5 < 5 * 16; t5 += NT) {   // (a workgroup may be one wave: LDPC_HIP_BP_LDS_THREADS=64)
            const uint32_t d = t5 >> 4, pat = t5 & 15;
            double v[4], sum = 0.0, post = prior;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) v[k] = (pat >> k) & 1u ? -v0 : v0;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++)
                if (k < d) {
                    sum += v[k];
                    post += v[k];
                }
            tab_p[d * 16 + pat] = post;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++)
                if (k < d) tab_t[((d - 1) * 16 + pat) * 4 + k] = bp_tanh_half<BPL_SK>(sum - v[k] + prior);
        }
    }
    if (tid == 0)
        for (uint32_t i = 0; i < 5; i++) cnt[i] = 0;
    // A run that never converges ends up repeating itself: on the degenerate quantum codes every such run alternates
    // between two states from about its tenth iteration on.  The update is a function of the bit messages alone, so
    // once they equal, bit for bit, those of two iterations before, the run is periodic for good: its stop tests keep
    // failing as they did in the last two iterations, and the decisions after max_iter iterations are the current
    // ones or the next ones, by parity.  From the sixth iteration on the even iterations leave their bit messages in
    // `saved` (global, this workgroup's) and compare with what is there; a periodic run takes at most one more
    // iteration and reports max_iter.  Results and counters are those of the full run (tests/test_gpu_bp_lds.py).
    unsigned long long *saved = a.msg ? (unsigned long long *)a.msg + (size_t)blockIdx.x * (2 * (size_t)E - EL) : nullptr;
    double *gmsg = HYB ? (double *)(saved + E) : nullptr;
    auto ldm = [&](uint32_t e) -> double { return !HYB || e < EL ? msg[e] : gmsg[e - EL]; };
    auto stm = [&](uint32_t e, double v) {
        if (!HYB || e < EL) msg[e] = v;
        else gmsg[e - EL] = v;
    };
    // (the index of the next shot is asked for at the start of the current one's first bit update and put in the
    // other queue slot at its end: the queue's round trip is off the path and the answer is not kept across a check row)
    if (tid == 0) sh_queue[0] = atomicAdd(a.queue, 1ull);
    // HYB: the checks and the bits of a phase are handed to the waves 64 at a time from a counter in LDS instead of by
    // thread index.  A check or bit whose messages live in the L2 piece costs about twice one whose messages are in
    // LDS, the fixed split gave some waves one such check in five and others two, and every phase ends at a barrier:
    // wave 0 spent a sixth of a shot waiting there (profiles/r02_bp_lds_phases.txt).  One counter serves all phases:
    // it only grows, a phase's chunks are its values from `wbase` on, and a phase takes exactly chunks + waves of them
    // (every wave's last grab fails), which every wave adds to its own copy of wbase -- no reset, no extra barrier.
    // Each check and bit is computed by the same code whichever wave takes it: results do not depend on the hand-out.
    // LISTED: the set bits of a shot's error are listed (u16 positions) by the threads that make its words, and the
    // syndrome phase works the list -- one (error bit, check) pair per thread, one round -- instead of testing every bit
    // of the word.  The list and its counter borrow the start of the message array, which is dead between a shot's
    // last iteration and its first bit update (the first iteration comes from the table and writes every message).
    constexpr bool LISTED = BPL_LISTED_ERRORS && FIRST_BY_TABLE && (HYB || BPL_LISTED_ERRORS > 1);
    uint32_t *lcnt = (uint32_t *)msg;
    uint16_t *llist = (uint16_t *)(lcnt + 1);
    // (a code with hardly any edges -- most bits in no check -- has a message array shorter than a full list: it keeps
    // the scan of the word)
    const bool listed = LISTED && (size_t)EL * 8 >= 4 + 2 * (size_t)n;
    if (listed && tid == 0) *lcnt = 0;
    uint32_t *sh_work = q + 67;   // (the spare word after the gap table's 65, sh_flag and sh_fail)
    uint32_t wbase = 0;
    if (HYB && tid == 0) *sh_work = 0;
    // (a wave asks for its next chunk before it works the current one, so the counter's round trip is off the path:
    // ask() leaves the answer in lane 0, taken() reads it when the chunk is started)
    auto ask = [&]() -> uint32_t {
        uint32_t v = 0;
        if (bpl_lane() == 0) v = atomicAdd(sh_work, 1u);
        return v;
    };
    auto taken = [&](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane(v) - wbase; };
#ifdef BPL_PROF
    uint64_t prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#endif
    for (uint32_t turn = 0;; turn ^= 1u) {
        __syncthreads();   // (also: the last shot's failure test has read its words, the tables and graph copies are written)
        const uint64_t myi_v = sh_queue[turn];
        const uint64_t myi = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(myi_v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)myi_v);
        if (myi >= a.n_shots) break;
        BPL_STAMP(0);
        bool asked = false;
        const uint64_t shot = a.first_shot + myi;
        {
        BPL_FRESH();
        if (tid == 0) *sh_fail = 0;
        // ---- the shot's error word (or the caller's), decision words, messages = the prior (:60-68)
        for (uint32_t w = tid; w < Wn; w += NT) {
            const uint32_t nb = bits_in_word(n, w);
            uint64_t x = 0;
            if (a.source == SRC_BSC) {
#if BPL_WORDS_BY_BLOCK
                // (the word block by block with the four gaps of a block looked up side by side and without branches:
                // the same stream and the same word as bsc_word; a look-up that the float estimate misses -- or that
                // reads the flag word behind the table at j = 64 -- takes the exact search)
                if (!a.gap.p_is_zero) {
                    uint32_t pos = 0;
                    for (uint32_t blk = 0; !bsc_block(q, a.gap.inv_log2q, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)shot, (uint32_t)(shot >> 32), w,
                                                      blk, nb, x, pos);
                         blk++) {
                    }
                }
#else
                x = bsc_word(q, a.gap.inv_log2q, a.gap.p_is_zero, a.seed, shot, w, nb);
#endif
            } else if (a.source == SRC_WORDS) {
                x = a.words_in[myi * Wn + w];
                if (nb < 64) x &= (1ull << nb) - 1;
            } else if (a.source != SRC_SYNDROMES) {
                uint64_t xx, zz;
                depol_word(q, a.gap.inv_log2q, a.gap.p_is_zero, a.seed, shot, w, nb, xx, zz);
                x = a.source == SRC_DEPOL_X ? xx : zz;
            }
            ew[w] = x;
            if (listed && a.source != SRC_SYNDROMES && x) {
                uint16_t *lp = llist + atomicAdd(lcnt, (uint32_t)__builtin_popcountll(x));
                for (uint64_t y = x; y; y &= y - 1) *lp++ = (uint16_t)(w * 64 + (uint32_t)__builtin_ctzll(y));
            }
            dw[w] = init ? (nb < 64 ? (1ull << nb) - 1 : ~0ull) : 0ull;
        }
        // (syndrome words: the caller's, or zero for the scatter below; the residual starts as syndrome ^ H * (decision
        // before the first update), where H * (all ones) is the parity of the check's degree)
        for (uint32_t w = tid; w < Wm; w += NT) {
            const uint32_t nb = bits_in_word(m, w);
            const uint64_t valid = nb < 64 ? (1ull << nb) - 1 : ~0ull;
            const uint64_t x = a.source == SRC_SYNDROMES ? a.synd_in[myi * Wm + w] & valid : 0ull;
            sw[w] = x;
            rw[w] = x ^ (init && DC > 0 && (DC & 1) ? valid : 0ull);
        }
        if (!FIRST_BY_TABLE)
            for (uint32_t e = tid; e < E; e += NT) stm(e, t_prior);
        }
        __syncthreads();
        BPL_STAMP(1);
        {
        BPL_FRESH();
        // ---- syndrome H e (linear_code/mod.rs:393-405) and the stop test's residual syndrome ^ H * decision: every set
        // bit of the error toggles its checks (few bits are set: E gathers become weight * degree atomics in LDS).  The
        // bit update keeps the residual current from then on.
        if (listed && a.source != SRC_SYNDROMES) {
            const uint32_t n_listed = *lcnt;
            if constexpr (DV > 0) {
                for (uint32_t t = tid; t < n_listed * (uint32_t)DV; t += NT) {
                    const uint32_t c = check_of((uint32_t)llist[t / (uint32_t)DV] * (uint32_t)DV + t % (uint32_t)DV);
                    atomicXor(&sw32[c >> 5], 1u << (c & 31));
                    atomicXor(&rw32[c >> 5], 1u << (c & 31));
                }
            } else {
                for (uint32_t t = tid; t < n_listed; t += NT) {
                    const uint32_t b = llist[t];
                    if constexpr (DV == -4 && BPL_TPACK_AHEAD > 1) {   // (the bit's checks in one 16-byte row)
                        const uint4 r = g.t_pack4[b];
                        const uint32_t w4[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++)
                            if (w4[k] != NONE) {
                                const uint32_t c = w4[k] >> 16;
                                atomicXor(&sw32[c >> 5], 1u << (c & 31));
                                atomicXor(&rw32[c >> 5], 1u << (c & 31));
                            }
                    } else {
                        for (uint32_t k = first_edge(b); k < first_edge(b + 1); k++) {
                            const uint32_t c = check_of(k);
                            atomicXor(&sw32[c >> 5], 1u << (c & 31));
                            atomicXor(&rw32[c >> 5], 1u << (c & 31));
                        }
                    }
                }
            }
        } else if (a.source != SRC_SYNDROMES)
            for (uint32_t b = tid; b < n; b += NT)
                if ((ew32[b >> 5] >> (b & 31)) & 1u) {
                    const uint32_t k0 = DV > 0 ? b * (uint32_t)DV : first_edge(b), k1 = DV > 0 ? k0 + (uint32_t)DV : first_edge(b + 1);
                    for (uint32_t k = k0; k < k1; k++) {
                        const uint32_t c = check_of(k);
                        atomicXor(&sw32[c >> 5], 1u << (c & 31));
                        atomicXor(&rw32[c >> 5], 1u << (c & 31));
                    }
                }
        if (DC == 0 && init) {   // (checks of mixed degree and a negative prior: the degrees' parities, 64 checks per wave and trip)
            __syncthreads();
            for (uint32_t c0 = wv * 64; c0 < m; c0 += NT) {
                const uint32_t c = c0 + lane;
                const uint64_t odd = __ballot(c < m && ((g.h_ptr[c + 1] - g.h_ptr[c]) & 1u));
                if (lane == 0) rw[c0 >> 6] ^= odd;
            }
        }
        }
        uint32_t iter = 0, ran = 0;
        bool conv, compared = false, last_one = false;
        BPL_STAMP(2);
        for (;;) {
            // ---- stop test (belief_propagation.rs:26-27): H * decision == syndrome, i.e. the residual is zero (every
            // wave reads all of it: no flag to share)
            __syncthreads();
            {
                BPL_FRESH();
                (void)tid;
                uint32_t r = 0;
                for (uint32_t w = lane; w < 2 * Wm; w += 64) r |= rw32[w];
                conv = __ballot(r != 0) == 0;
            }
            bool periodic = false;
            if (compared) {   // (wave-uniform, the same in every wave: a second barrier before the flag is reused)
                periodic = *sh_flag == 0;
                __syncthreads();
            }
            BPL_STAMP(3);
            if (conv || iter >= a.max_iter) break;
            if (last_one || (periodic && !((a.max_iter - iter) & 1u))) {
                iter = a.max_iter;   // the run would go on alternating between this state and the last one
                break;
            }
            last_one = periodic;
            const uint32_t t = iter + 1;
            const bool save = saved != nullptr && t >= BPL_PERIOD_FROM && !(t & 1u) && !last_one;
            compared = save && t >= BPL_PERIOD_FROM + 2;
            const bool by_table = FIRST_BY_TABLE && iter == 0;
            {
            BPL_FRESH();
            (void)lane;
            if (compared && tid == 0) *sh_flag = 0;
            // ---- update_checks (:150-166), in place: tanh of the bit messages in, check messages out
            if constexpr (HYB && DC > 0) {
                if (!by_table) {
                    // (dealing the chunks whose messages live in the L2 piece between the others, instead of last, made no
                    // difference beyond the 1 % from run to run)
                    const uint32_t c_chunks = (m + 63) / 64;
                    uint32_t vcur = ask();
                    for (;;) {
                        const uint32_t ch = taken(vcur);
                        if (ch >= c_chunks) break;
                        vcur = ask();
                        const uint32_t c = ch * 64 + lane;
                        if (c < m) {
                            const bool sbit = (sw32[c >> 5] >> (c & 31)) & 1u;
                            double x[DC], out[DC];
#pragma unroll
                            for (int j = 0; j < DC; j++) x[j] = ldm(c * (uint32_t)DC + j);
                            bp_check_row<DC, BPL_CHECK_BATCH, BPL_SK>(x, sbit ? 0x80000000u : 0u, out);
#pragma unroll
                            for (int j = 0; j < DC; j++) stm(c * (uint32_t)DC + j, out[j]);
                        }
                    }
                    wbase += c_chunks + NT / 64;
                }
            } else {
            for (uint32_t c = tid; c < m && !by_table; c += NT) {
                const bool sbit = (sw32[c >> 5] >> (c & 31)) & 1u;
                if constexpr (DC > 0) {
                    double x[DC], out[DC];
#pragma unroll
                    for (int j = 0; j < DC; j++) x[j] = ldm(c * (uint32_t)DC + j);
                    bp_check_row<DC, BPL_CHECK_BATCH, BPL_SK>(x, sbit ? 0x80000000u : 0u, out);
#pragma unroll
                    for (int j = 0; j < DC; j++) stm(c * (uint32_t)DC + j, out[j]);
                } else {
                    // checks of any degree: bp_kernel's general row, operation for operation
                    const uint32_t e0 = g.h_ptr[c], dc = g.h_ptr[c + 1] - e0;
                    if (dc <= REG_DEG) {
                        double t[REG_DEG];
                        double prod = 1.0;
#pragma unroll
                        for (uint32_t j = 0; j < REG_DEG; j++)
                            if (j < dc) {
                                t[j] = ldm(e0 + j);
                                prod *= t[j];
                            }
#pragma unroll
                        for (uint32_t j = 0; j < REG_DEG; j++)
                            if (j < dc) {
                                double v = 2.0 * bp_ref_atanh<BPL_SK>(bp_div_prod(prod, t[j]));
                                if (sbit) v *= -1.0;
                                stm(e0 + j, v);
                            }
                    } else {
                        double prod = 1.0;
                        for (uint32_t j = 0; j < dc; j++) prod *= ldm(e0 + j);
                        for (uint32_t j = 0; j < dc; j++) {
                            double v = 2.0 * bp_ref_atanh<BPL_SK>(bp_div_prod(prod, ldm(e0 + j)));
                            if (sbit) v *= -1.0;
                            stm(e0 + j, v);
                        }
                    }
                }
            
            }
            }
            }
            __syncthreads();
            BPL_STAMP(4);
            {
            BPL_FRESH();
            // ---- update_bits (:168-180) + hard decision (:100-113), in place: check messages in, tanh((sum - own + prior) / 2)
            // out; 64 consecutive bits per wave and trip, so a ballot is the decision word
            unsigned long long next_i = 0;
            if (!asked && tid == 0) next_i = atomicAdd(a.queue, 1ull);
            const uint32_t b_chunks = (n + 63) / 64;
            // AHEAD (bits of one fixed degree): a bit's packed graph words (edge | check << 16, one per check) are asked
            // for one chunk ahead, from global memory whether or not LDS holds copies of the graph: the loads of chunk
            // i + 1 travel while chunk i is worked, instead of a memory (or LDS) round trip at the head of every chunk.
            // HYB keeps two hand-outs in flight for that (the chunk after next is asked for while the next one's words
            // are loaded): every wave's last TWO grabs fail, which is what wbase counts.
            // (bits of mixed degree up to four, DV < 0: the same from the rows of t_pack4, 16 bytes per bit; the bit's
            // degree is the number of its words that are not the filler)
            constexpr bool AHEAD = (BPL_TPACK_AHEAD && DV > 0) || (BPL_TPACK_AHEAD > 1 && DV == -4);
            constexpr uint32_t TPN = DV > 0 ? (uint32_t)DV : DV == -4 ? 4u : 1u;
            uint32_t tp[TPN], tpn[TPN];
            auto load_tp = [&](uint32_t at, uint32_t (&t)[TPN]) {
                if constexpr (DV > 0) {
#pragma unroll
                    for (uint32_t k = 0; k < TPN; k++) t[k] = at + lane < n ? g.t_pack[(at + lane) * TPN + k] : 0u;
                } else if constexpr (DV == -4) {
                    uint4 r = make_uint4(NONE, NONE, NONE, NONE);
                    if (at + lane < n) r = g.t_pack4[at + lane];
                    t[0] = r.x; t[1] = r.y; t[2] = r.z; t[3] = r.w;
                }
            };
            uint32_t vcur = 0, b0 = wv * 64, b0n = 0;
            if (HYB) {
                const uint32_t v1 = ask();
                vcur = ask();
                b0 = taken(v1) * 64;
            }
            if (AHEAD) load_tp(b0, tp);
            for (; b0 < n; b0 = b0n) {
                if (HYB) {
                    b0n = taken(vcur) * 64;
                    vcur = ask();
                } else b0n = b0 + NT;
                if (AHEAD) load_tp(b0n, tpn);
                const uint32_t b = b0 + lane;
                bool dec = false, changed = false;
                if (b < n) {
                    constexpr uint32_t RD = DV > 0 ? (uint32_t)DV : DV < 0 ? (uint32_t)(-DV) : REG_DEG;
                    uint32_t k0 = 0, dv = 0;
                    if constexpr (DV > 0) {
                        k0 = b * (uint32_t)DV;
                        dv = (uint32_t)DV;
                    } else if constexpr (AHEAD) {
#pragma unroll
                        for (uint32_t k = 0; k < TPN; k++) dv += tp[k] != NONE ? 1u : 0u;
                    } else {
                        k0 = first_edge(b);
                        dv = first_edge(b + 1) - k0;
                    }
                    double sum = 0.0, post = prior;
                    if (by_table) {
                        uint32_t pat = 0, pk[RD];
#pragma unroll
                        for (uint32_t k = 0; k < RD; k++)
                            if (DV > 0 || k < dv) pk[k] = AHEAD ? tp[k < TPN ? k : 0] : slot_of(k0 + k);
#pragma unroll
                        for (uint32_t k = 0; k < RD; k++)
                            if (DV > 0 || k < dv) {
                                const uint32_t c = pk[k] >> 16;
                                pat |= ((sw32[c >> 5] >> (c & 31)) & 1u) << k;
                            }
#pragma unroll
                        for (uint32_t k = 0; k < RD; k++)
                            if (DV > 0 || k < dv) stm(pk[k] & 0xFFFFu, tab_t[((dv - 1) * 16 + pat) * 4 + k]);
                        post = tab_p[dv * 16 + pat];
                    } else if (DV != 0 || dv <= RD) {
                        uint32_t ed[RD];
                        double v[RD], y[RD];
#pragma unroll
                        for (uint32_t k = 0; k < RD; k++) {
                            v[k] = 0.0;
                            if (DV > 0 || k < dv) {
                                ed[k] = AHEAD ? tp[k < TPN ? k : 0] & 0xFFFFu : edge_of(k0 + k);
                                v[k] = ldm(ed[k]);
                            }
                        }
#pragma unroll
                        for (uint32_t k = 0; k < RD; k++)
                            if (DV > 0 || k < dv) {
                                sum += v[k];
                                post += v[k];
                            }
#pragma unroll
                        for (uint32_t k = 0; k < RD; k++) v[k] = sum - v[k] + prior;
                        if constexpr (DV != 0 && BPL_TANH_BATCH == 2) {
                            double v2[2], y2[2];
                            v2[0] = v[0]; v2[1] = v[1];
                            bp_tanh_half_n<2, BPL_SK>(v2, y2, 2);
                            y[0] = y2[0]; y[1] = y2[1];
                            v2[0] = v[2]; v2[1] = RD > 3 ? v[RD - 1] : 0.0;
                            if (DV < 0 && __ballot(dv == RD) != 0) bp_tanh_half_n<2, BPL_SK>(v2, y2, 2);
                            else bp_tanh_half_n<2, BPL_SK>(v2, y2, 1);
                            y[2] = y2[0];
                            if (RD > 3) y[RD - 1] = y2[1];
                        } else if constexpr (DV != 0 && BPL_TANH_BATCH > 2) {
                            // (the tanh of a bit's three or four messages as one batch: the polynomial's constants
                            // are built once; the last of the batch only when some bit of the wave has that many checks)
                            if (DV > 0 || __ballot(dv == RD) != 0) bp_tanh_half_n<(int)RD, BPL_SK>(v, y, (int)RD);
                            else bp_tanh_half_n<(int)RD, BPL_SK>(v, y, (int)RD - 1);
                        } else {
#ifdef BPL_TANH_BATCH_ANY   // (diagnostic build: round 2's dropped variant, the batch for bits of any degree up to REG_DEG)
                            bp_tanh_half_n<(int)RD, BPL_SK>(v, y, BPL_TANH_BATCH_ANY == 1 ? (int)dv : (int)__builtin_amdgcn_readfirstlane(wave_max_u32(dv)));
#else
#pragma unroll
                            for (uint32_t k = 0; k < RD; k++)
                                if (DV > 0 || k < dv) y[k] = bp_tanh_half<BPL_SK>(v[k]);
#endif
                        }
#pragma unroll
                        for (uint32_t k = 0; k < RD; k++)
                            if (DV > 0 || k < dv) stm(ed[k], y[k]);
                    } else {
                        for (uint32_t k = 0; k < dv; k++) {
                            const double vv = ldm(edge_of(k0 + k));
                            sum += vv;
                            post += vv;
                        }
                        for (uint32_t k = 0; k < dv; k++) {
                            const uint32_t e = edge_of(k0 + k);
                            stm(e, bp_tanh_half<BPL_SK>(sum - ldm(e) + prior));
                        }
                    }
                    dec = post < 0.0;
                    changed = dec != (((dw32[b >> 5] >> (b & 31)) & 1u) != 0);
                    if (changed) {   // (few bits after the first iteration)
                        if constexpr (AHEAD) {
#pragma unroll
                            for (uint32_t k = 0; k < TPN; k++)
                                if (DV > 0 || k < dv) {
                                    const uint32_t c = tp[k] >> 16;
                                    atomicXor(&rw32[c >> 5], 1u << (c & 31));
                                }
                        } else {
                            for (uint32_t k = 0; k < dv; k++) {
                                const uint32_t c = check_of(k0 + k);
                                atomicXor(&rw32[c >> 5], 1u << (c & 31));
                            }
                        }
                    }
                }
                const uint64_t mask = __ballot(dec);
                if (lane == 0) dw[b0 >> 6] = mask;
                if (AHEAD) {
#pragma unroll
                    for (uint32_t k = 0; k < TPN; k++) tp[k] = tpn[k];
                }
            }
            if (HYB) wbase += b_chunks + 2 * (NT / 64);
            if (!asked) {
                if (tid == 0) sh_queue[turn ^ 1u] = next_i;
                asked = true;
            }
            }
            BPL_STAMP(by_table ? 5 : 6);
            if (save) {   // (wave-uniform; only runs that are still going after five iterations come here)
                __syncthreads();
                BPL_FRESH();
                bool differs = false;
                for (uint32_t e = tid; e < E; e += NT) {
                    const unsigned long long nb = (unsigned long long)__double_as_longlong(ldm(e));
                    if (compared && saved[e] != nb) differs = true;
                    saved[e] = nb;
                }
                if (compared && __ballot(differs) && lane == 0) *sh_flag = 1;
            }
            BPL_STAMP(7);
            iter++;
            if (!by_table) ran++;   // (iterations of message arithmetic: what rooflines are priced on)
        }
        // ---- the shot has stopped: failure test, outputs (css/mod.rs:225-252 for the logical test)
        BPL_FRESH();
        if (listed && tid == 0) *lcnt = 0;   // (the messages are dead: the next shot's error list starts here)
        if (tid == 0 && !asked) sh_queue[turn ^ 1u] = atomicAdd(a.queue, 1ull);   // (a shot that stopped before its first update)
        const bool det = !conv;
        bool fail = false;
        if (a.test != TEST_NONE) {   // (every wave left the loop behind its barrier: the words are final)
            uint32_t f = 0;
            if (a.test == TEST_RESIDUAL) {
                for (uint32_t w = tid; w < Wn; w += NT) f |= (ew[w] ^ dw[w]) != 0 ? 1u : 0u;
            } else if (!det) {   // (a run that did not converge has failed already, css.rs / css_decoding.rs:18)
                // eight lanes per logical generator: its entries come in a few strided loads instead of one long chain
                for (uint32_t r0 = 0; r0 < a.l_rows; r0 += NT / 8) {
                    const uint32_t r = r0 + (tid >> 3);
                    uint32_t par = 0;
                    if (r < a.l_rows) {
                        const uint32_t k1 = c_logicals ? (uint32_t)lptr[r + 1] : a.l_ptr[r + 1];
                        for (uint32_t k = (c_logicals ? (uint32_t)lptr[r] : a.l_ptr[r]) + (tid & 7u); k < k1; k += 8) {
                            const uint32_t b = c_logicals ? (uint32_t)lidx[k] : a.l_idx[k];
                            par ^= ((ew32[b >> 5] ^ dw32[b >> 5]) >> (b & 31)) & 1u;
                        }
                    }
                    par ^= (uint32_t)__shfl_xor((int)par, 1);
                    par ^= (uint32_t)__shfl_xor((int)par, 2);
                    par ^= (uint32_t)__shfl_xor((int)par, 4);
                    f |= par;
                }
            }
            if (__ballot(f != 0) && lane == 0) *sh_fail = 1;
            BPL_STAMP(8);
            __syncthreads();
            BPL_STAMP(9);
            fail = a.test == TEST_RESIDUAL ? *sh_fail != 0 : det || *sh_fail != 0;
        }
        if (a.corr_out)
            for (uint32_t w = tid; w < Wn; w += NT) a.corr_out[myi * Wn + w] = a.xor_input ? dw[w] ^ ew[w] : dw[w];
        if (tid == 0) {
            if (a.iters_out) a.iters_out[myi] = iter;
            if (a.failed_out && fail) a.failed_out[myi] |= (uint8_t)a.fail_bit;
            cnt[0]++; cnt[1] += fail; cnt[2] += det; cnt[3] += iter; cnt[4] += ran;
        }
        BPL_STAMP(8);
    }
    const unsigned long long c_shots = tid == 0 ? cnt[0] : 0, c_fail = tid == 0 ? cnt[1] : 0, c_det = tid == 0 ? cnt[2] : 0,
                             c_work = tid == 0 ? cnt[3] : 0, c_ran = tid == 0 ? cnt[4] : 0;
#ifdef BPL_PROF
    if (blockIdx.x == 0 && tid == 0 && c_shots)
        printf("bp_lds phases, cycles per shot over %llu shots (%llu iterations by arithmetic): queue %.0f  words %.0f  syndrome %.0f  "
               "stop %.0f  checks %.0f  bits(table) %.0f  bits %.0f  save/compare %.0f  harvest %.0f (of which its barrier %.0f)\n", (unsigned long long)c_shots,
               (unsigned long long)c_ran, (double)prof[0] / c_shots, (double)prof[1] / c_shots, (double)prof[2] / c_shots, (double)prof[3] / c_shots,
               (double)prof[4] / c_shots, (double)prof[5] / c_shots, (double)prof[6] / c_shots, (double)prof[7] / c_shots, (double)(prof[8] + prof[9]) / c_shots, (double)prof[9] / c_shots);
#endif
    if (tid == 0 && a.executed && c_ran) atomicAdd(a.executed, (unsigned long long)c_ran);
    if (tid == 0 && a.counters) {
        if (c_shots) atomicAdd(&a.counters[CNT_SHOTS], (unsigned long long)c_shots);
        if (c_fail) atomicAdd(&a.counters[CNT_FAIL], (unsigned long long)c_fail);
        if (c_det) {
            atomicAdd(&a.counters[CNT_DET], (unsigned long long)c_det);
            atomicAdd(&a.counters[CNT_CAPPED], (unsigned long long)c_det);
        }
        if (c_fail > c_det && a.test == TEST_RESIDUAL) atomicAdd(&a.counters[CNT_UNDET], (unsigned long long)(c_fail - c_det));
        if (c_work) atomicAdd(&a.counters[CNT_WORK], (unsigned long long)c_work);
    }
}

// ---------------------------------------------------------------------------
// Belief propagation, streamed form for codes whose message arrays are so large
// that memory allows fewer 64-shot groups than the chip has room for workgroups
// (C5: n = 10^6, 3 GB per group).  The same data layout and arithmetic as
// bp_kernel, but every phase is a grid-wide kernel over (group, part): all CUs
// work on every group, kernel boundaries are the barriers, the slot state lives
// in global memory (BpsState) and the host only reads one counter per round.
// ---------------------------------------------------------------------------
constexpr uint32_t BPS_THREADS = 256, BPS_WAVES = BPS_THREADS / 64;

struct BpsGroup {
    double *bitsm, *chkm;
    uint64_t *ew, *dw, *dsl, *ssl, *esl, *psl;
};

__device__ inline BpsGroup bps_group(const DeviceGraph &g, const BpArgs &a, size_t grp)
{
    const size_t Wn = (g.n + 63) / 64;
    BpsGroup r;
    r.bitsm = a.msg + grp * 2 * (size_t)g.E * 64;
    r.chkm = r.bitsm + (size_t)g.E * 64;
    r.ew = a.ew + grp * Wn * 64;
    r.dw = a.dw + grp * Wn * 64;
    r.dsl = a.dsl + grp * (size_t)g.n;
    r.ssl = a.ssl + grp * (size_t)g.m;
    r.esl = a.esl + grp * (size_t)g.n;
    r.psl = a.psl + grp * (size_t)g.m;
    return r;
}

__device__ inline size_t bps_grp(const BpsState &s, uint32_t i) { return s.gmap ? s.gmap[i] : i; }

// slots whose shot stops now: converged (stop test) or at the iteration cap (belief_propagation.rs:26-27)
__device__ inline uint64_t bps_done_mask(const BpArgs &a, const BpsState &s, size_t grp, uint32_t lane, bool &conv)
{
    const bool active = (s.act[grp] >> lane) & 1ull;
    conv = !((s.mism[grp] >> lane) & 1ull);
    return __ballot(active && (conv || s.iter[grp * 64 + lane] >= a.max_iter));
}

// one wave per group: hand idle slots their next shots
__global__ __launch_bounds__(64) void bps_ctl_refill(BpArgs a, BpsState s, uint32_t g0)
{
    const size_t grp = bps_grp(s, g0 + blockIdx.x);
    const uint32_t lane = threadIdx.x;
    const uint64_t act = s.act[grp], idle = ~act;
    uint64_t first = a.n_shots;
    if (idle) {
        unsigned long long f = 0;
        if (lane == 0) f = atomicAdd(a.queue, (unsigned long long)__builtin_popcountll(idle));
        first = readlane64(f, 0);
    }
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0));
    const bool take = ((idle >> lane) & 1ull) && first < a.n_shots && first + rank < a.n_shots;
    const uint64_t takemask = __ballot(take);
    if (take) {
        s.myi[grp * 64 + lane] = first + rank;
        s.iter[grp * 64 + lane] = 0;
    }
    s.flags[grp * 64 + lane] = 0;
    if (lane == 0) {
        s.take[grp] = takemask;
        s.act[grp] = act | takemask;
        s.fresh[grp] = (s.fresh[grp] & act) | takemask;
        s.mism[grp] = 0;
    }
}

// (group, part): error words of the taken slots, bit-sliced rows, initial decisions
__global__ __launch_bounds__(BPS_THREADS) void bps_setup_words(DeviceGraph g, BpArgs a, BpsState s, uint32_t parts, uint32_t g0)
{
    __shared__ uint32_t q[65];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t grp = bps_grp(s, g0 + blockIdx.x / parts);
    const uint32_t part = blockIdx.x % parts;
    const uint64_t takemask = s.take[grp];
    if (!takemask) return;
    for (uint32_t i = tid; i < 65; i += BPS_THREADS) q[i] = a.gap.q[i];
    __syncthreads();
    const BpsGroup G = bps_group(g, a, grp);
    const uint32_t n = g.n, Wn = (n + 63) / 64, Wm = (g.m + 63) / 64;
    const bool take = (takemask >> lane) & 1ull;
    const uint64_t myi = take ? s.myi[grp * 64 + lane] : 0, shot = a.first_shot + myi;
    const bool init = a.prior < 0.0;   // decision before the first update: posterior = prior
    for (uint32_t w = part * BPS_WAVES + wv; w < Wn; w += parts * BPS_WAVES) {
        const uint32_t nb = bits_in_word(n, w);
        const uint32_t b = w * 64 + lane;
        if (a.source != SRC_SYNDROMES) {
            uint64_t x = 0;
            if (take) {
                if (a.source == SRC_BSC) x = bsc_word(q, a.gap.inv_log2q, a.gap.p_is_zero, a.seed, shot, w, nb);
                else if (a.source == SRC_WORDS) {
                    x = a.words_in[myi * Wn + w];
                    if (nb < 64) x &= (1ull << nb) - 1;
                } else {
                    uint64_t xx, zz;
                    depol_word(q, a.gap.inv_log2q, a.gap.p_is_zero, a.seed, shot, w, nb, xx, zz);
                    x = a.source == SRC_DEPOL_X ? xx : zz;
                }
                G.ew[(size_t)w * 64 + lane] = x;
            }
            const uint64_t anyv = wave_or64(x);
            uint64_t any = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(anyv >> 32)) << 32) |
                           __builtin_amdgcn_readfirstlane((uint32_t)anyv);
            uint64_t row = 0;
            while (any) {
                const uint32_t i = (uint32_t)__builtin_ctzll(any);
                any &= any - 1;
                const uint64_t mask = __ballot((x >> i) & 1ull);
                if (lane == i) row = mask;
            }
            if (b < n) G.esl[b] = row;
        }
        if (b < n) G.dsl[b] = (G.dsl[b] & ~takemask) | (init ? takemask : 0ull);
        if (take) G.dw[(size_t)w * 64 + lane] = init ? (nb < 64 ? (1ull << nb) - 1 : ~0ull) : 0ull;
    }
    if (a.source == SRC_SYNDROMES) {
        // given syndromes: the same ballot transpose, word by word, straight into the bit-sliced rows
        for (uint32_t w = part * BPS_WAVES + wv; w < Wm; w += parts * BPS_WAVES) {
            uint64_t x = take ? a.synd_in[myi * Wm + w] : 0;
            const uint64_t anyv = wave_or64(x);
            uint64_t any = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(anyv >> 32)) << 32) |
                           __builtin_amdgcn_readfirstlane((uint32_t)anyv);
            uint64_t row = 0;
            while (any) {
                const uint32_t i = (uint32_t)__builtin_ctzll(any);
                any &= any - 1;
                const uint64_t mask = __ballot((x >> i) & 1ull);
                if (lane == i) row = mask;
            }
            const uint32_t c = w * 64 + lane;
            if (c < g.m) {
                G.ssl[c] = (G.ssl[c] & ~takemask) | (row & takemask);
                const uint64_t ipar = init && ((g.h_ptr[c + 1] - g.h_ptr[c]) & 1u) ? ~0ull : 0ull;   // H * (all-ones decision)
                G.psl[c] = (G.psl[c] & ~takemask) | ((row ^ ipar) & takemask);
            }
        }
    }
}

// (group, part): syndromes of the taken slots from the bit-sliced error rows, thread per check
__global__ __launch_bounds__(BPS_THREADS) void bps_setup_synd(DeviceGraph g, BpArgs a, BpsState s, uint32_t parts, uint32_t g0)
{
    const size_t grp = bps_grp(s, g0 + blockIdx.x / parts);
    const uint32_t part = blockIdx.x % parts;
    const uint64_t takemask = s.take[grp];
    if (!takemask || a.source == SRC_SYNDROMES) return;
    const BpsGroup G = bps_group(g, a, grp);
    for (uint32_t c = part * BPS_THREADS + threadIdx.x; c < g.m; c += parts * BPS_THREADS) {
        uint64_t par = 0;
        for (uint32_t e = g.h_ptr[c]; e < g.h_ptr[c + 1]; e++) par ^= G.esl[g.h_idx[e]];
        G.ssl[c] = (G.ssl[c] & ~takemask) | (par & takemask);
        // residual of the decision before the first update (bps_setup_words: all ones when the prior is negative)
        const uint64_t ipar = a.prior < 0.0 && ((g.h_ptr[c + 1] - g.h_ptr[c]) & 1u) ? ~0ull : 0ull;
        G.psl[c] = (G.psl[c] & ~takemask) | ((par ^ ipar) & takemask);
    }
}

// (group, part): residual rows of every check from the decision rows (the persistent kernel's tail arrives without them)
__global__ __launch_bounds__(BPS_THREADS) void bps_parity_init(DeviceGraph g, BpArgs a, BpsState s, uint32_t parts, uint32_t g0)
{
    const size_t grp = bps_grp(s, g0 + blockIdx.x / parts);
    const uint32_t part = blockIdx.x % parts;
    if (!s.act[grp]) return;
    const BpsGroup G = bps_group(g, a, grp);
    for (uint32_t c = part * BPS_THREADS + threadIdx.x; c < g.m; c += parts * BPS_THREADS) {
        uint64_t x = G.ssl[c];
        for (uint32_t e = g.h_ptr[c]; e < g.h_ptr[c + 1]; e++) x ^= G.dsl[g.h_idx[e]];
        G.psl[c] = x;
    }
}

// (group, part): stop test, bit-sliced; only_fresh: groups that took no shot this round are skipped
__global__ __launch_bounds__(BPS_THREADS) void bps_stop(DeviceGraph g, BpArgs a, BpsState s, uint32_t parts, uint32_t g0, int only_fresh)
{
    const size_t grp = bps_grp(s, g0 + blockIdx.x / parts);
    const uint32_t part = blockIdx.x % parts;
    if (!s.act[grp] || (only_fresh && !s.take[grp])) return;
    const BpsGroup G = bps_group(g, a, grp);
    uint64_t acc = 0;
    // (the residual rows are kept up to date by the bit update: one coalesced pass, no gathers)
    for (uint32_t c = part * BPS_THREADS + threadIdx.x; c < g.m; c += parts * BPS_THREADS) acc |= G.psl[c];
    acc = wave_or64(acc);
    if ((threadIdx.x & 63) == 0 && acc) atomicOr((unsigned long long *)&s.mism[grp], (unsigned long long)acc);
}

// (group, part): outputs and residual test of the slots that stop now
__global__ __launch_bounds__(BPS_THREADS) void bps_harvest_words(DeviceGraph g, BpArgs a, BpsState s, uint32_t parts, uint32_t g0, int only_fresh)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t grp = bps_grp(s, g0 + blockIdx.x / parts);
    const uint32_t part = blockIdx.x % parts;
    if (!s.act[grp] || (only_fresh && !s.take[grp])) return;
    bool conv;
    const uint64_t donemask = bps_done_mask(a, s, grp, lane, conv);
    if (!donemask) return;
    const bool done = (donemask >> lane) & 1ull;
    const BpsGroup G = bps_group(g, a, grp);
    const uint32_t n = g.n, Wn = (n + 63) / 64;
    const uint64_t myi = done ? s.myi[grp * 64 + lane] : 0;
    const bool fresh = (s.fresh[grp] >> lane) & 1ull;
    uint64_t rany = 0;
    for (uint32_t w = part * BPS_WAVES + wv; w < Wn; w += parts * BPS_WAVES) {
        if (!done) continue;
        const uint64_t d = G.dw[(size_t)w * 64 + lane];
        if (a.source != SRC_SYNDROMES) {
            const uint64_t e = G.ew[(size_t)w * 64 + lane];
            rany |= e ^ d;
            if (a.corr_out) a.corr_out[myi * Wn + w] = a.xor_input ? d ^ e : d;
        } else if (a.corr_out) a.corr_out[myi * Wn + w] = d;
    }
    if (a.test == TEST_RESIDUAL && rany) atomicOr(&s.flags[grp * 64 + lane], 1u);
    if (a.test == TEST_LOGICAL) {
        // CSS half (css/mod.rs:225-252): a logical generator that anticommutes with the residual
        uint32_t lfail = 0;
        for (uint32_t r = part * BPS_WAVES + wv; r < a.l_rows; r += parts * BPS_WAVES) {
            uint32_t par = 0;
            for (uint32_t k = a.l_ptr[r]; k < a.l_ptr[r + 1]; k++) {
                const uint32_t b = a.l_idx[k];
                if (done) {
                    const size_t off = (size_t)(b >> 6) * 64 + lane;
                    par ^= (uint32_t)((G.ew[off] ^ G.dw[off]) >> (b & 63)) & 1u;
                }
            }
            lfail |= par;
        }
        if (lfail) atomicOr(&s.flags[grp * 64 + lane], 2u);
    }
    if (a.post_out && done)
        for (uint32_t b = part * BPS_WAVES + wv; b < n; b += parts * BPS_WAVES) {      // BpState::decode posterior, :100-106
            double l = a.prior;
            if (!fresh)
                for (uint32_t k = g.t_ptr[b]; k < g.t_ptr[b + 1]; k++) l += G.chkm[(size_t)g.t_edge[k] * 64 + lane];
            a.post_out[myi * n + b] = l;
        }
    if (a.chk_out && done)
        for (uint32_t e = part * BPS_WAVES + wv; e < g.E; e += parts * BPS_WAVES)
            a.chk_out[myi * g.E + e] = fresh ? 0.0 : G.chkm[(size_t)e * 64 + lane];
}

// one wave per group: counters and per-shot outputs of the slots that stop, slots freed; alive: add the
// slots still active (plus one while the shot queue holds shots) to *alive, the host's loop condition
__global__ __launch_bounds__(64) void bps_ctl_harvest(BpArgs a, BpsState s, uint32_t g0, int only_fresh, unsigned long long *alive)
{
    const size_t grp = bps_grp(s, g0 + blockIdx.x);
    const uint32_t lane = threadIdx.x;
    uint64_t act = s.act[grp];
    if (act && !(only_fresh && !s.take[grp])) {
        bool conv;
        const uint64_t donemask = bps_done_mask(a, s, grp, lane, conv);
        const bool done = (donemask >> lane) & 1ull;
        const bool det = !conv;
        const uint32_t fl = s.flags[grp * 64 + lane];
        const bool fail = a.test == TEST_RESIDUAL ? fl != 0 : a.test == TEST_LOGICAL ? det || (fl & 2u) != 0 : false;
        const uint32_t iter = s.iter[grp * 64 + lane];
        if (done) {
            const uint64_t myi = s.myi[grp * 64 + lane];
            if (a.iters_out) a.iters_out[myi] = iter;
            if (a.failed_out && fail) a.failed_out[myi] |= (uint8_t)a.fail_bit;
        }
        const uint64_t c_shots = __builtin_popcountll(donemask), c_fail = __builtin_popcountll(__ballot(done && fail)),
                       c_det = __builtin_popcountll(__ballot(done && det)), c_work = wave_sum64(done ? iter : 0);
        if (lane == 0 && donemask && a.counters) {
            atomicAdd(&a.counters[CNT_SHOTS], (unsigned long long)c_shots);
            if (c_fail) atomicAdd(&a.counters[CNT_FAIL], (unsigned long long)c_fail);
            if (c_det) {
                atomicAdd(&a.counters[CNT_DET], (unsigned long long)c_det);
                atomicAdd(&a.counters[CNT_CAPPED], (unsigned long long)c_det);
            }
            if (c_fail > c_det && a.test == TEST_RESIDUAL) atomicAdd(&a.counters[CNT_UNDET], (unsigned long long)(c_fail - c_det));
            if (c_work) atomicAdd(&a.counters[CNT_WORK], (unsigned long long)c_work);
        }
        act &= ~donemask;
        if (lane == 0) s.act[grp] = act;
    }
    if (alive && lane == 0 && act) atomicAdd(alive, (unsigned long long)__builtin_popcountll(act));
    // shots still queued keep the host's loop going even when every slot has just stopped (fresh shots with a zero
    // syndrome stop in the round that loads them)
    if (alive && lane == 0 && blockIdx.x == 0 && *a.queue < a.n_shots) atomicAdd(alive, 1ull);
}

// (group, part): update_checks
template <int DC>
__device__ inline void bps_checks_body(const DeviceGraph &g, const BpArgs &a, const BpsState &s, size_t grp, uint32_t part, uint32_t parts)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t actmask = s.act[grp];
    if (!actmask) return;
    const BpsGroup G = bps_group(g, a, grp);
    const bool active = (actmask >> lane) & 1ull, fresh = (s.fresh[grp] >> lane) & 1ull;
    const double t_prior = bp_tanh_half(a.prior);
    const uint32_t m = g.m, first = part * BPS_WAVES + wv, step = parts * BPS_WAVES;
    if constexpr (DC > 0) {
        double xn[DC];
        if (first < m) {
#pragma unroll
            for (int j = 0; j < DC; j++) xn[j] = fresh ? t_prior : G.bitsm[((size_t)first * DC + j) * 64 + lane];
        }
        for (uint32_t c = first; c < m; c += step) {
            const size_t e0 = (size_t)c * DC;
            const bool sbit = (G.ssl[c] >> lane) & 1ull;
            double x[DC];
#pragma unroll
            for (int j = 0; j < DC; j++) x[j] = xn[j];
            if (c + step < m) {
#pragma unroll
                for (int j = 0; j < DC; j++) xn[j] = fresh ? t_prior : G.bitsm[((size_t)(c + step) * DC + j) * 64 + lane];
            }
            double out[DC];
            bp_check_row<DC>(x, sbit ? 0x80000000u : 0u, out);
#pragma unroll
            for (int j = 0; j < DC; j++)
                if (active) G.chkm[(e0 + j) * 64 + lane] = out[j];
        }
    } else {
        for (uint32_t c = first; c < m; c += step) {
            const uint32_t e0 = g.h_ptr[c], dc = g.h_ptr[c + 1] - e0;
            const bool sbit = (G.ssl[c] >> lane) & 1ull;
            double prod = 1.0;
            for (uint32_t j = 0; j < dc; j++) prod *= fresh ? t_prior : G.bitsm[(size_t)(e0 + j) * 64 + lane];
            for (uint32_t j = 0; j < dc; j++) {
                const double t = fresh ? t_prior : G.bitsm[(size_t)(e0 + j) * 64 + lane];
                double v = 2.0 * bp_ref_atanh(bp_div_prod(prod, t));
                if (sbit) v *= -1.0;
                if (active) G.chkm[(size_t)(e0 + j) * 64 + lane] = v;
            }
        }
    }
}

// (group, part): update_bits + hard decisions
template <int DV>
__device__ inline void bps_bits_body(const DeviceGraph &g, const BpArgs &a, const BpsState &s, size_t grp, uint32_t part, uint32_t parts,
                                     uint32_t split)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t actmask = s.act[grp];
    if (!actmask) return;
    const BpsGroup G = bps_group(g, a, grp);
    const bool active = (actmask >> lane) & 1ull;
    const uint32_t Wn = (g.n + 63) / 64;
    // (split: pieces per 64-bit chunk, 1 when the groups in play fill the chip with whole chunks; see bp_stream_run)
    for (uint32_t item = part * BPS_WAVES + wv; item < Wn * split; item += parts * BPS_WAVES)
        bp_bits_chunk<DV>(g, item, split, lane, active, actmask, a.prior, G.chkm, G.bitsm, G.dw, G.dsl, G.psl);
}

// One half-step of the streamed form: update_checks on the groups [gc0, gc0 + gcn) and update_bits on the
// groups [gb0, gb0 + gbn), blocks of the two roles interleaved so that every CU holds both: the check update
// is bound by f64 issue, the bit update by HBM, and two group sets half an iteration apart keep both busy.
template <int DC, int DV>
__global__ __launch_bounds__(BPS_THREADS, 4) void bps_iterate(DeviceGraph g, BpArgs a, BpsState s, uint32_t parts, uint32_t gc0,
                                                              uint32_t gcn, uint32_t gb0, uint32_t gbn, uint32_t split)
{
    const uint32_t nc = gcn * parts, nb = gbn * parts, mn = nc < nb ? nc : nb;
    bool checks;
    uint32_t idx;
    if (blockIdx.x < 2 * mn) {
        checks = !(blockIdx.x & 1);
        idx = blockIdx.x >> 1;
    } else {
        checks = nc > nb;
        idx = blockIdx.x - mn;
    }
    if (checks) bps_checks_body<DC>(g, a, s, bps_grp(s, gc0 + idx / parts), idx % parts, parts);
    else bps_bits_body<DV>(g, a, s, bps_grp(s, gb0 + idx / parts), idx % parts, parts, split);
}

// one wave per group: the iteration is over
__global__ __launch_bounds__(64) void bps_ctl_advance(BpsState s, uint32_t g0)
{
    const size_t grp = bps_grp(s, g0 + blockIdx.x);
    const uint32_t lane = threadIdx.x;
    const uint64_t act = s.act[grp];
    if ((act >> lane) & 1ull) s.iter[grp * 64 + lane] += 1;
    if (lane == 0) {
        s.fresh[grp] &= ~act;
        s.mism[grp] = 0;
    }
}

// (destination group, part): tail compaction.  Once nothing is left to refill, the shots still running are
// spread thinly over many groups, and a group costs a full pass over its message rows however few of its 64
// slots are active.  The host lists moves (moves[j * 64 + slot] = source group * 64 + source slot, or
// 0xFFFFFFFF) from the active slots of the groups that are given up into free slots of the groups that are kept;
// this kernel copies a moved shot's state: its column of the bit-message rows (the check messages too when
// posteriors are wanted: at an iteration boundary nothing else reads them), of the error / decision words, its
// bit of the bit-sliced decision and syndrome rows, its shot index and iteration count.  Sources are active
// slots, destinations free ones: no move reads what another writes.  Lanes compute nothing across slots, so
// a shot's results do not depend on where it sits.
__global__ __launch_bounds__(BPS_THREADS) void bps_compact(DeviceGraph g, BpArgs a, BpsState s, uint32_t parts, const uint32_t *moves,
                                                           int with_chk)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t j = blockIdx.x / parts, part = blockIdx.x % parts;
    const size_t dgrp = bps_grp(s, j);
    const uint32_t mv = moves[(size_t)j * 64 + lane];
    const bool recv = mv != 0xFFFFFFFFu;
    const uint64_t recvmask = __ballot(recv);
    if (!recvmask) return;
    const size_t sgrp = recv ? (size_t)(mv >> 6) : dgrp;
    const uint32_t sslot = recv ? (mv & 63u) : lane;
    const BpsGroup D = bps_group(g, a, dgrp), S = bps_group(g, a, sgrp);
    const uint32_t n = g.n, m = g.m, E = g.E, Wn = (n + 63) / 64;
    const uint32_t nwv = parts * BPS_WAVES, w0 = part * BPS_WAVES + wv;
    for (uint32_t e = w0; e < E; e += nwv)
        if (recv) {
            D.bitsm[(size_t)e * 64 + lane] = S.bitsm[(size_t)e * 64 + sslot];
            if (with_chk) D.chkm[(size_t)e * 64 + lane] = S.chkm[(size_t)e * 64 + sslot];
        }
    for (uint32_t w = w0; w < Wn; w += nwv)
        if (recv) {
            D.ew[(size_t)w * 64 + lane] = S.ew[(size_t)w * 64 + sslot];
            D.dw[(size_t)w * 64 + lane] = S.dw[(size_t)w * 64 + sslot];
        }
    // bit-sliced rows, 64 at a time: lane i ends up with the row of bit (check) b0 + i
    for (uint32_t b0 = w0 * 64; b0 < n; b0 += nwv * 64) {
        const uint32_t cnt = n - b0 < 64u ? n - b0 : 64u;
        uint64_t row = 0;
        for (uint32_t i = 0; i < cnt; i++) {
            const bool bit = recv && ((S.dsl[b0 + i] >> sslot) & 1ull);
            const uint64_t mask = __ballot(bit);
            if (lane == i) row = mask;
        }
        if (lane < cnt) D.dsl[b0 + lane] = (D.dsl[b0 + lane] & ~recvmask) | row;
    }
    for (uint32_t c0 = w0 * 64; c0 < m; c0 += nwv * 64) {
        const uint32_t cnt = m - c0 < 64u ? m - c0 : 64u;
        uint64_t row = 0;
        for (uint32_t i = 0; i < cnt; i++) {
            const bool bit = recv && ((S.ssl[c0 + i] >> sslot) & 1ull);
            const uint64_t mask = __ballot(bit);
            if (lane == i) row = mask;
        }
        if (lane < cnt) D.ssl[c0 + lane] = (D.ssl[c0 + lane] & ~recvmask) | row;
        row = 0;
        for (uint32_t i = 0; i < cnt; i++) {
            const bool bit = recv && ((S.psl[c0 + i] >> sslot) & 1ull);
            const uint64_t mask = __ballot(bit);
            if (lane == i) row = mask;
        }
        if (lane < cnt) D.psl[c0 + lane] = (D.psl[c0 + lane] & ~recvmask) | row;
    }
    if (part == 0 && wv == 0 && recv) {
        s.iter[dgrp * 64 + lane] = s.iter[mv];
        s.myi[dgrp * 64 + lane] = s.myi[mv];
        s.flags[dgrp * 64 + lane] = 0;   // (the harvest ORs into it; the slot's last shot may have left bits)
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static uint32_t grid_for(uint64_t items, uint32_t per_block, uint32_t cap)
{
    uint64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (uint32_t)(b < cap ? b : cap);
}

size_t bp_workspace_bytes_per_group(const DeviceGraph &g)
{
    size_t Wn = (g.n + 63) / 64;
    return 2 * (size_t)g.E * 64 * sizeof(double) + 2 * Wn * 64 * sizeof(uint64_t) + (2 * (size_t)g.n + 2 * (size_t)g.m) * sizeof(uint64_t);
}

hipError_t launch_sample_bsc(const GapTable &gap, uint32_t n, uint64_t seed, uint64_t first, uint64_t n_shots,
                             uint64_t *d_words, hipStream_t st)
{
    uint64_t total = n_shots * ((n + 63) / 64);
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(sample_bsc_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, st, gap, n, seed, first,
                       n_shots, d_words);
    return hipGetLastError();
}

hipError_t launch_sample_depol(const GapTable &gap, uint32_t n, uint64_t seed, uint64_t first, uint64_t n_shots,
                               uint64_t *d_x, uint64_t *d_z, hipStream_t st)
{
    uint64_t total = n_shots * ((n + 63) / 64);
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(sample_depol_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, st, gap, n, seed, first,
                       n_shots, d_x, d_z);
    return hipGetLastError();
}

hipError_t launch_syndrome(const DeviceGraph &g, uint64_t n_shots, const uint64_t *d_words, uint64_t *d_synd,
                           hipStream_t st)
{
    if (!n_shots || !g.m) return hipSuccess;
    const size_t per_wave = (size_t)((g.m + 63) / 64) * 8;
    uint32_t nw = 4;
    while (nw > 1 && nw * per_wave > 48 * 1024) nw >>= 1;
    if (nw * per_wave <= 64 * 1024)
        hipLaunchKernelGGL(syndrome_sparse_kernel, dim3(grid_for(n_shots, nw, 16384)), dim3(64 * nw), nw * per_wave, st, g,
                           n_shots, d_words, d_synd);
    else
        hipLaunchKernelGGL(syndrome_kernel, dim3(grid_for(n_shots, 4, 8192)), dim3(256), 0, st, g, n_shots, d_words, d_synd);
    return hipGetLastError();
}

hipError_t launch_flip(const DeviceGraph &g, const FlipArgs &a, hipStream_t st)
{
    if (!a.n_shots) return hipSuccess;
    size_t Wn = (g.n + 63) / 64, Wm = (g.m + 63) / 64;
    size_t per_wave = (Wm + 2 * Wn) * 8;
    uint32_t wpb = 4;
    while (wpb > 1 && 33 * 8 + wpb * per_wave > 64 * 1024) wpb >>= 1;
    size_t lds = 33 * 8 + wpb * per_wave;
    if (lds > 160 * 1024) return hipErrorInvalidValue;   // caller reports LDPC_ERR_CAPACITY
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)flip_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    g_kernel_name = "flip_kernel";
    hipLaunchKernelGGL(flip_kernel, dim3(grid_for(a.n_shots, wpb, 256 * 8)), dim3(wpb * 64), lds, st, g, a);
    return hipGetLastError();
}

// waves (= shots in flight) of a flip_big_kernel launch, and the scratch they index
// (bound by memory round trips: the more shots in flight the better.  32 waves per CU = 8192 slices of 312 KB at C5:
// 1.0e4 shots/s against 9.0e3 with 16 per CU on calls of 8192 shots and more; the kernel is held to 64 registers for it)
uint32_t flip_big_waves(uint64_t n_shots) { return (uint32_t)std::min<uint64_t>(n_shots, (uint64_t)device_cu_count() * 32); }
size_t flip_big_scratch_bytes(const DeviceGraph &g, uint64_t n_shots)
{
    return ((size_t)flip_big_waves(n_shots) + 3) / 4 * 4 * flip_big_words(g.n, g.m) * 8;
}

hipError_t launch_flip_big(const DeviceGraph &g, const FlipArgs &a, uint64_t *big, hipStream_t st)
{
    if (!a.n_shots) return hipSuccess;
    const uint32_t blocks = (flip_big_waves(a.n_shots) + 3) / 4;
    g_kernel_name = "flip_big_kernel";
    hipLaunchKernelGGL(flip_big_kernel, dim3(blocks), dim3(256), 0, st, g, a, big);
    return hipGetLastError();
}

size_t flip2_lds_bytes(const DeviceGraph &g, uint32_t lanes_per_shot, uint32_t list_cap)
{
    size_t Wn = (g.n + 63) / 64, Wm = (g.m + 63) / 64;
    return (flip2_hdr_words(list_cap) + (64 / lanes_per_shot) * (Wm + Wn + (2 * Wn + 63) / 64)) * 8;
}

uint32_t flip2_lanes_per_shot(const DeviceGraph &g)
{
    // 8 lanes per shot (8 shots per wave) halves the instructions per shot; needs 8 states in LDS
    // (measured at C2: 16 lanes x 4 shots beats 8 x 8: one round per flip, twice the waves in flight)
    return (uint32_t)switches().flip_lanes;   // A/B aid (LDPC_HIP_FLIP_LANES)
}

// list capacity for an expected error weight: the next power of two above mean + 4 sigma, within [64, 1024]
uint32_t flip2_list_cap(double expected_weight)
{
    const double want = expected_weight + 4.0 * std::sqrt(expected_weight > 0 ? expected_weight : 0.0);
    uint32_t cap = 64;
    while (cap < 1024 && (double)cap < want) cap <<= 1;
    return cap;
}

static uint32_t flip2_blocks_for(const DeviceGraph &g, uint32_t list_cap)
{
    // one wave per block; blocks per CU limited by LDS (160 KiB) and by 8 waves per SIMD
    const uint32_t lanes = flip2_lanes_per_shot(g);
    return device_cu_count() * (uint32_t)std::min<size_t>(32, (160 * 1024) / flip2_lds_bytes(g, lanes, list_cap));
}

// upper bound on the blocks of any launch (scratch sizing): the smallest list
uint32_t flip2_max_blocks(const DeviceGraph &g) { return flip2_blocks_for(g, 64); }

template <int DVP, int G>
static void launch_flip2_t(const DeviceGraph &g, const NbTable &nb, const FlipArgs &a, uint32_t grid, size_t lds, hipStream_t st)
{
    const bool diag = a.max_flips != 0xFFFFFFFFu;
    const bool one = !diag && (1u << nb.kshift) == (uint32_t)G;
    const bool small = one && (2 * ((g.n + 63) / 64) + 31) / 32 <= (uint32_t)G;
#define LDPC_FLIP2(REG_, DIAG_, ONE_, SMALL_) \
    hipLaunchKernelGGL((flip2_kernel<DVP, G, REG_, DIAG_, ONE_, SMALL_>), dim3(grid), dim3(64), lds, st, g, nb, a)
    if (nb.regular) {
        if (diag) LDPC_FLIP2(true, true, false, false);
        else if (small) LDPC_FLIP2(true, false, true, true);
        else if (one) LDPC_FLIP2(true, false, true, false);
        else LDPC_FLIP2(true, false, false, false);
    } else {
        if (diag) LDPC_FLIP2(false, true, false, false);
        else if (small) LDPC_FLIP2(false, false, true, true);
        else if (one) LDPC_FLIP2(false, false, true, false);
        else LDPC_FLIP2(false, false, false, false);
    }
#undef LDPC_FLIP2
}

hipError_t launch_flip2(const DeviceGraph &g, const NbTable &nb, const FlipArgs &a, hipStream_t st, uint32_t block_cap)
{
    if (!a.n_shots) return hipSuccess;
    const uint32_t lanes = flip2_lanes_per_shot(g);
    const size_t lds = flip2_lds_bytes(g, lanes, a.list_cap);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    uint32_t grid = grid_for(a.n_shots, 64 / lanes, flip2_blocks_for(g, a.list_cap));
    if (block_cap) grid = std::min(grid, block_cap);
    g_kernel_name = "flip2_kernel";
    if (nb.dvp == 3) {
        if (lanes == 8) launch_flip2_t<3, 8>(g, nb, a, grid, lds, st);
        else launch_flip2_t<3, 16>(g, nb, a, grid, lds, st);
    } else {
        if (lanes == 8) launch_flip2_t<7, 8>(g, nb, a, grid, lds, st);
        else launch_flip2_t<7, 16>(g, nb, a, grid, lds, st);
    }
    return hipGetLastError();
}

namespace {
Switches g_switches;
std::once_flag g_switches_once;
Switches read_switches()
{
    Switches s;
    if (const char *e = std::getenv("LDPC_HIP_FLIP_MAXFLIPS")) s.flip_maxflips = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char *e = std::getenv("LDPC_HIP_FLIP_LANES")) s.flip_lanes = std::atoi(e) == 8 ? 8 : 16;
    s.flip_prof = std::getenv("LDPC_HIP_FLIP_PROF") != nullptr;
    if (const char *e = std::getenv("LDPC_HIP_FLIP_BPC")) s.flip_bpc = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("LDPC_HIP_FLIP_NF")) s.flip_nf = std::atoi(e);
    if (const char *e = std::getenv("LDPC_HIP_FLIP_KERNEL")) s.flip_kernel = !std::strcmp(e, "csr") ? 1 : !std::strcmp(e, "v2") ? 2 : !std::strcmp(e, "v3") ? 3 : 0;
    if (const char *e = std::getenv("LDPC_HIP_BP_WAVES")) s.bp_waves = std::max(1, std::min(8, std::atoi(e)));
    if (const char *e = std::getenv("LDPC_HIP_BP_OVERSUB")) s.bp_oversub = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("LDPC_HIP_BP_PATH")) s.bp_path = !std::strcmp(e, "stream") ? 1 : !std::strcmp(e, "persistent") ? 2 : !std::strcmp(e, "lds") ? 3 : !std::strcmp(e, "hbm") ? 4 : 0;
    if (const char *e = std::getenv("LDPC_HIP_BP_GROUPS")) s.bp_groups = std::max(1ll, std::atoll(e));
    if (const char *e = std::getenv("LDPC_HIP_BP_HANDOFF")) {
        s.bp_handoff_set = true;
        s.bp_handoff = (uint32_t)std::strtoul(e, nullptr, 10);
    }
    s.bp_profile = std::getenv("LDPC_HIP_BP_PROFILE") != nullptr;
    if (const char *e = std::getenv("LDPC_HIP_BP_HYB_KB")) s.bp_hyb_kb = (uint32_t)std::max(8, std::min(158, std::atoi(e)));
    if (const char *e = std::getenv("LDPC_HIP_BP_LDS_THREADS")) s.bp_lds_threads = (uint32_t)std::max(64, std::min(1024, std::atoi(e))) / 64 * 64;
    s.bp_stream_skew = std::getenv("LDPC_HIP_BP_STREAM_SKEW") != nullptr;
    if (const char *e = std::getenv("LDPC_HIP_BP_TAIL_BURST")) s.bp_tail_burst = std::max(1, std::atoi(e));
    s.bp_no_compact = std::getenv("LDPC_HIP_BP_NO_COMPACT") != nullptr;
    if (const char *e = std::getenv("LDPC_HIP_CSS_SIDE_BY_SIDE")) s.css_side_by_side = std::atoi(e) != 0 ? 1 : 0;
    s.debug = std::getenv("LDPC_HIP_DEBUG") != nullptr;
    return s;
}
}  // namespace

const char *last_launched_kernel() { return g_kernel_name; }

const Switches &switches()
{
    std::call_once(g_switches_once, [] { g_switches = read_switches(); });
    return g_switches;
}

void reload_switches()
{
    (void)switches();
    g_switches = read_switches();
}

// multiprocessors of the current device (cached per device index)
uint32_t device_cu_count()
{
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] <= 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached[dev] = cus;
    }
    return (uint32_t)cached[dev];
}

size_t flip3_lds_bytes(const Nb3Table &t, uint32_t list_cap)
{
    return ((size_t)flip3_hdr_words(list_cap) + 8 * ((size_t)t.sw + (size_t)t.windows * 8)) * 8;
}

// list capacity for an expected error weight: mean + 4 sigma, in steps of 64 within [64, 1024]
uint32_t flip3_list_cap(double expected_weight)
{
    const double want = expected_weight + 4.0 * std::sqrt(expected_weight > 0 ? expected_weight : 0.0) + 8.0;
    uint32_t cap = 64;
    while (cap < 1024 && (double)cap < want) cap += 64;
    return cap;
}

// upper bound on the blocks of any flip3 launch (scratch sizing): 8 one-wave blocks per SIMD
uint32_t flip3_max_blocks() { return device_cu_count() * 32u; }

// ---------------------------------------------------------------------------
// Scratch of the entry point that allocates nothing (ldpc_flip_mc_run_noalloc): a file-scope device array, one copy
// per GPU for the life of the process.  In host code `scratch` is only a stand-in for the symbol: its device address
// on the current GPU comes from hipGetSymbolAddress, once per device, and everything (memset, kernel arguments,
// copies) uses that pointer.
// ---------------------------------------------------------------------------
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)64 << 20;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

hipError_t noalloc_scratch(unsigned char **ptr, size_t *bytes)
{
    static unsigned char *per_device[64];
    static std::mutex mu;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (!per_device[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));   // this GPU's copy of the array (no allocation)
        if (e != hipSuccess) return e;
        per_device[dev] = (unsigned char *)p;
    }
    *ptr = per_device[dev];
    *bytes = NOALLOC_SCRATCH_BYTES;
    return hipSuccess;
}

hipError_t launch_flip3(const DeviceGraph &g, const Nb3Table &nb, const FlipArgs &a, bool prof, hipStream_t st, uint32_t block_cap)
{
    if (!a.n_shots) return hipSuccess;
    const size_t lds = flip3_lds_bytes(nb, a.list_cap);
    if (lds > 64 * 1024 || a.list_cap % 64 || a.list_cap < 64) return hipErrorInvalidValue;
    uint32_t per_cu = (uint32_t)std::min<size_t>(32, (160 * 1024) / lds);
    if (switches().flip_bpc) per_cu = std::min<uint32_t>(per_cu, (uint32_t)switches().flip_bpc);
    if (switches().debug) {
        int occ = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)flip3_kernel<false>, 64, lds);
        std::fprintf(stderr, "[ldpc_hip] flip3: lds %zu B, %u blocks per CU (occupancy API: %d)\n", lds, per_cu, occ);
    }
    uint32_t grid = grid_for(a.n_shots, 8, device_cu_count() * per_cu);
    if (block_cap) grid = std::min(grid, block_cap);   // (the caller's scratch holds fewer blocks: each works more batches)
    g_kernel_name = "flip3_kernel";
    if (prof) hipLaunchKernelGGL(flip3_kernel<true>, dim3(grid), dim3(64), lds, st, g, nb, a);
    else hipLaunchKernelGGL(flip3_kernel<false>, dim3(grid), dim3(64), lds, st, g, nb, a);
    return hipGetLastError();
}

// flip4: slots that fit the CU's LDS beside the header (a little is left unused: the hardware's allocation granule)
uint32_t flip4_slots(const Nb3Table &t, uint32_t list_cap, uint32_t nf)
{
    const size_t hdr = (size_t)flip4_hdr_words(list_cap, 16 - nf) * 8, per = ((size_t)t.sw + (size_t)t.windows * 8) * 8;
    const size_t room = 160 * 1024 - 2048;
    if (hdr + per > room) return 0;
    return (uint32_t)std::min<size_t>(F4_MAX_SLOTS, (room - hdr) / per);
}

uint32_t flip4_scratch_pieces() { return device_cu_count() * F4_LOGS; }

typedef void (*flip4_fn)(DeviceGraph, Nb3Table, FlipArgs, uint32_t);
static flip4_fn flip4_kernel_for(uint32_t nf, bool prof)
{
    // (SV = 4: the hand-written loop of passes; 3: its compiled form flip3_step_v2; 0: round 2's flip3_step.  Codes
    // 100 * SV' + flip waves are A/B aids: LDPC_HIP_FLIP_NF.)
    switch (nf) {
    case 4: return flip4_kernel<4, false, 8, 12, 4>;
    case 6: return flip4_kernel<6, false, 8, 10, 4>;
    case 7: return prof ? flip4_kernel<7, true, 8, 9, 4> : flip4_kernel<7, false, 8, 9, 4>;
    case 8: return prof ? flip4_kernel<8, true, 8, 8, 4> : flip4_kernel<8, false, 8, 8, 4>;
    case 9: return prof ? flip4_kernel<9, true, 8, 7, 4> : flip4_kernel<9, false, 8, 7, 4>;
    case 10: return flip4_kernel<10, false, 8, 6, 4>;
    case 109: return prof ? flip4_kernel<9, true> : flip4_kernel<9>;
    case 108: return flip4_kernel<8>;
    case 119: return flip4_kernel<9, false, 8, 7, 1>;   // issue-slot probes on round 2's pass: ten more scalar / vector instructions
    case 129: return flip4_kernel<9, false, 8, 7, 2>;
    case 309: return flip4_kernel<9, false, 8, 7, 3>;
    case 308: return flip4_kernel<8, false, 8, 8, 3>;
    default: return nullptr;
    }
}

hipError_t launch_flip4(const DeviceGraph &g, const Nb3Table &nb, const FlipArgs &a, hipStream_t st)
{
    if (!a.n_shots) return hipSuccess;
    if (a.list_cap % 64 || a.list_cap < 64) return hipErrorInvalidValue;
    // flip waves of the 16: measured at C2 8 -> 7.2e7, 9 -> 7.6e7 shots/s, 10 -> the service waves do not keep up
    // (LDPC_HIP_FLIP_NF: A/B aid); fewer when the LDS holds fewer slots than their groups plus one per service wave
    uint32_t nf = switches().flip_nf >= 4 && switches().flip_nf <= 10 ? (uint32_t)switches().flip_nf : 8u;
    while (nf > 4 && (!flip4_kernel_for(nf, false) || flip4_slots(nb, a.list_cap, nf) < nf * 8 + (16 - nf))) nf--;
    flip4_fn k = flip4_kernel_for(nf, switches().flip_prof);
    if (switches().flip_nf > 100 && flip4_kernel_for(switches().flip_nf, false)) {   // A/B forms: 100 * variant + flip waves
        nf = (uint32_t)switches().flip_nf % 100;
        k = flip4_kernel_for(switches().flip_nf, switches().flip_prof);
    }
    const uint32_t n_slots = flip4_slots(nb, a.list_cap, nf);
    if (!k || n_slots < nf * 8 + 1) return hipErrorInvalidValue;
    const size_t lds = ((size_t)flip4_hdr_words(a.list_cap, 16 - nf) + (size_t)n_slots * (nb.sw + nb.windows * 8)) * 8;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(device_cu_count(), (a.n_shots + 63) / 64);
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (switches().debug) std::fprintf(stderr, "[ldpc_hip] flip4: %u flip waves, %u slots, lds %zu B, grid %u\n", nf, n_slots, lds, grid);
    static const char *const names[] = {"flip4_kernel<4>", "flip4_kernel<5>", "flip4_kernel<6>", "flip4_kernel<7>", "flip4_kernel<8>",
                                        "flip4_kernel<9>", "flip4_kernel<10>"};
    g_kernel_name = names[nf - 4];
    hipLaunchKernelGGL(k, dim3(grid), dim3(1024), lds, st, g, nb, a, n_slots);
    return hipGetLastError();
}

typedef void (*bp_kernel_fn)(DeviceGraph, BpArgs);

static bp_kernel_fn bp_kernel_for(int dc, int dv)
{
    if (dc == 6 && dv == 3) return bp_kernel<6, 3>;
    if (dc == 4 && dv == 3) return bp_kernel<4, 3>;
    // dv = -4: bits of mixed degree, none above 4; a regular bit degree up to 4 without a kernel of its own is served
    // by the same instantiation
    const bool small_dv = dv == -4 || (dv >= 1 && dv <= 4);
    if (dc == 7 && small_dv) return bp_kernel<7, -4>;
    if (dc == 7) return bp_kernel<7, 0>;
    if (small_dv) return bp_kernel<0, -4>;
    return bp_kernel<0, 0>;
}

// workgroups of `waves` wavefronts that are resident at once on the whole device (asked of the runtime once per
// device and kernel form: the single-vector calls take 0.1-0.4 ms and should not pay an occupancy query each)
uint32_t bp_resident_groups(uint32_t waves, int dc, int dv)
{
    static std::mutex mu;
    static std::map<std::tuple<int, uint32_t, const void *>, uint32_t> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const void *fn = (const void *)bp_kernel_for(dc, dv);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(std::make_tuple(dev, waves, fn));
    if (it != cache.end()) return it->second;
    int per_cu = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)(64 * waves), 0) != hipSuccess || per_cu < 1) per_cu = 1;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t r = (uint32_t)per_cu * (uint32_t)(cus > 0 ? cus : 256);
    cache[std::make_tuple(dev, waves, fn)] = r;
    return r;
}

hipError_t launch_bp(const DeviceGraph &g, const BpArgs &a, uint32_t groups, uint32_t waves, int dc, int dv,
                     hipStream_t st)
{
    if (!a.n_shots || !groups) return hipSuccess;
    g_kernel_name = dc == 6 && dv == 3 ? "bp_kernel<6,3>" : dc == 4 && dv == 3 ? "bp_kernel<4,3>" : dc == 7 ? "bp_kernel<7,*>" : "bp_kernel<0,*>";
    hipLaunchKernelGGL(bp_kernel_for(dc, dv), dim3(groups), dim3(64 * waves), 0, st, g, a);
    return hipGetLastError();
}

typedef void (*bp_lds_fn)(DeviceGraph, BpArgs, uint32_t, uint32_t);
template <bool WIDE> static bp_lds_fn bp_lds_kernel_of(int dc, int v)
{
    switch (dc) {
    case 3: return v == 3 ? bp_lds_kernel<3, 3, WIDE> : v == -4 ? bp_lds_kernel<3, -4, WIDE> : bp_lds_kernel<3, 0, WIDE>;
    case 4: return v == 3 ? bp_lds_kernel<4, 3, WIDE> : v == -4 ? bp_lds_kernel<4, -4, WIDE> : bp_lds_kernel<4, 0, WIDE>;
    case 6: return v == 3 ? bp_lds_kernel<6, 3, WIDE> : v == -4 ? bp_lds_kernel<6, -4, WIDE> : bp_lds_kernel<6, 0, WIDE>;
    case 7: return v == 3 ? bp_lds_kernel<7, 3, WIDE> : v == -4 ? bp_lds_kernel<7, -4, WIDE> : bp_lds_kernel<7, 0, WIDE>;
    default: return v == -4 || v == 3 ? bp_lds_kernel<0, -4, WIDE> : bp_lds_kernel<0, 0, WIDE>;   // checks of any degree
    }
}
static bp_lds_fn bp_lds_kernel_for(int dc, int dv, bool wide = false)
{
    const int v = dv == 3 ? 3 : (dv == -4 || (dv >= 1 && dv <= 4)) ? -4 : 0;
    return wide ? bp_lds_kernel_of<true>(dc, v) : bp_lds_kernel_of<false>(dc, v);
}

// The hybrid form (one instantiation: check degree 6, bit degree 3, i.e. C2): a shot's messages beyond what LDS holds
// live in global memory.  e_lds = the edges kept in LDS (whole check rows).
static bp_lds_fn bp_lds_hybrid_kernel() { return bp_lds_kernel<6, 3, true, true>; }
static uint32_t bp_lds_hybrid_edges(const DeviceGraph &g)
{
    const size_t fixed = bp_lds_words(g, 0, false, 0, 0, 0) * 8, room = (size_t)(switches().bp_hyb_kb ? switches().bp_hyb_kb : 158) * 1024;
    return fixed >= room ? 0 : (uint32_t)std::min<size_t>((room - fixed) / 8 / 6 * 6, g.E);
}
bool bp_lds_hybrid_serves(const DeviceGraph &g, int dc, int dv)
{
    return dc == 6 && dv == 3 && g.E < 65536 && g.n < 65535 && g.t_pack && bp_lds_bytes(g) > 150 * 1024 && bp_lds_hybrid_edges(g) >= g.E / 8;
}
// ... and is ahead of bp_kernel: while LDS holds at least half of a shot's messages (the pieces in global memory stay in
// L2).  2^19 shots, (3,6)-regular, bp_kernel / hybrid ms: E = 24,000 292 / 198, 30,000 366 / 263, 36,000 441 / 358,
// 42,000 525 / 571, 48,000 605 / 873, 60,000 764 / 1456 (scripts/bp_hyb_reach.py).
bool bp_lds_hybrid_pays(const DeviceGraph &g, int dc, int dv) { return bp_lds_hybrid_serves(g, dc, dv) && 2 * (uint64_t)bp_lds_hybrid_edges(g) >= g.E; }

// smallest LDS footprint of a workgroup (no graph copies)
size_t bp_lds_bytes(const DeviceGraph &g) { return bp_lds_words(g, 0, false, 0, 0) * 8; }

bool bp_lds_serves(const DeviceGraph &g, int dc)
{
    return bp_lds_kernel_for(dc, 0) != nullptr && g.E > 0 && g.E < 65536 && g.n < 65535 && g.t_pack && bp_lds_bytes(g) <= 150 * 1024;   // (16-bit graph entries)
}

// The launch shape of the LDS form for a code: as many graph copies in LDS as cost no resident workgroup (asked of the
// runtime once per device, kernel form and footprint)
struct BpLdsShape {
    uint32_t copies, workgroups, threads;
    size_t lds;
};
static BpLdsShape bp_lds_shape(const DeviceGraph &g, int dc, int dv, uint32_t l_rows = 0, uint32_t l_nnz = 0)
{
    static std::mutex mu;
    static std::map<std::tuple<int, const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t>, BpLdsShape> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const void *fn = (const void *)bp_lds_kernel_for(dc, dv);
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_tuple(dev, fn, g.E, g.n, g.m, l_rows, l_nnz, switches().bp_lds_threads);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    const bool mixed = !(dv == 3);
    BpLdsShape best{0, 0, BPL_THREADS, 0};
    // (the most workgroups, then the most copies; the logical rows only when there are some and u16 holds their offsets)
    // (the forms that load a bit's packed graph words a chunk ahead -- every bit degree but "any" -- read nothing of the
    // graph from LDS: for them the choice is between no copy and the logical rows alone, level 4)
    const bool ahead = BPL_TPACK_AHEAD > 1 ? (dv == 3 || dv == -4 || (dv >= 1 && dv <= 4)) : BPL_TPACK_AHEAD == 1 && dv == 3;
    const uint32_t most = ahead ? (l_rows && l_nnz < 65536 ? 4 : 0) : l_rows && l_nnz < 65536 ? 3 : 2;
    for (uint32_t copies = 0; copies <= most; copies += ahead ? 4 : 1) {
        const size_t lds = bp_lds_words(g, copies, mixed, l_rows, l_nnz) * 8;
        if (lds > 160 * 1024) break;
        int per_cu = 0;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) break;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)BPL_THREADS, lds) != hipSuccess || per_cu < 1) per_cu = copies == 0 ? 1 : 0;
        const uint32_t wgs = (uint32_t)per_cu * device_cu_count();
        if (wgs && wgs >= best.workgroups) best = BpLdsShape{copies, wgs, BPL_THREADS, lds};
    }
    // Codes of which few shots fit a CU's LDS get more waves per shot: the workgroup size of 256, 512, 768 or 1024 threads
    // that puts the most threads on a CU with at most six waves per SIMD (seven measured slower, and the runtime's
    // occupancy answer for odd sizes was not what the hardware did), no wider than the code has checks or half its bits.
    // Measured on (3,6)-regular codes, 2^18 shots: E = 6000 512 threads 14.9 ms against 17.0 with 256, E = 9000 768
    // threads 24.8 against 33.7, E = 15,000 1024 threads 54 against 98 (scripts/bp_lds_threads.py).
    if (best.workgroups) {
        const uint32_t per_cu = best.workgroups / device_cu_count();
        const uint32_t useful = (std::max(g.m, (g.n + 1) / 2) + 63) / 64 * 64;
        uint32_t threads = BPL_THREADS;
        if (switches().bp_lds_threads) threads = switches().bp_lds_threads;
        else if (per_cu < 5) {
            uint32_t most_threads = per_cu * BPL_THREADS;
            for (uint32_t t = 512; t <= BPL_MAX_THREADS && t <= std::max(useful, BPL_THREADS); t += 256) {
                const uint32_t fit = std::min(per_cu, 24u / (t / 64));
                if (fit * t > most_threads) {
                    most_threads = fit * t;
                    threads = t;
                }
            }
        }
        if (threads != BPL_THREADS) {
            int fit = 0;
            const void *wide = (const void *)bp_lds_kernel_for(dc, dv, true);
            if (hipFuncSetAttribute(wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)best.lds) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, wide, (int)threads, best.lds) == hipSuccess && fit >= 1) {
                best.threads = threads;
                best.workgroups = std::min((uint32_t)fit, std::max(1u, switches().bp_lds_threads ? 28u / (threads / 64) : 24u / (threads / 64))) * device_cu_count();
            }
        }
    }
    cache[key] = best;
    return best;
}

static BpLdsShape bp_lds_hybrid_shape(const DeviceGraph &g)
{
    BpLdsShape sh{0, 0, switches().bp_lds_threads ? switches().bp_lds_threads : BPL_MAX_THREADS, bp_lds_words(g, 0, false, 0, 0, bp_lds_hybrid_edges(g)) * 8};
    int fit = 0;
    const void *fn = (const void *)bp_lds_hybrid_kernel();
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.lds) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, fn, (int)sh.threads, sh.lds) == hipSuccess && fit >= 1)
        sh.workgroups = std::min((uint32_t)fit, std::max(1u, 24u / (sh.threads / 64))) * device_cu_count();
    return sh;
}

uint32_t bp_lds_workgroups(const DeviceGraph &g, int dc, int dv)
{
    return bp_lds_serves(g, dc) ? bp_lds_shape(g, dc, dv).workgroups : bp_lds_hybrid_serves(g, dc, dv) ? bp_lds_hybrid_shape(g).workgroups : 0;
}

// scratch of the LDS form in a.msg: per workgroup the saved state of the period test (E words) and, hybrid form, the messages beyond LDS
size_t bp_lds_workspace_bytes(const DeviceGraph &g, int dc, int dv)
{
    const size_t el = bp_lds_serves(g, dc) ? g.E : bp_lds_hybrid_edges(g);
    return (size_t)bp_lds_workgroups(g, dc, dv) * (2 * (size_t)g.E - el) * 8;
}

hipError_t launch_bp_lds(const DeviceGraph &g, const BpArgs &a, int dc, int dv, hipStream_t st)
{
    if (!a.n_shots) return hipSuccess;
    if (!bp_lds_kernel_for(dc, dv) || a.post_out || a.chk_out) return hipErrorInvalidValue;
    const bool logical = a.test == TEST_LOGICAL && a.l_rows;
    const bool hybrid = !bp_lds_serves(g, dc);
    if (hybrid && !bp_lds_hybrid_serves(g, dc, dv)) return hipErrorInvalidValue;
    const BpLdsShape sh = hybrid ? bp_lds_hybrid_shape(g) : bp_lds_shape(g, dc, dv, logical ? a.l_rows : 0, logical ? a.l_nnz : 0);
    if (!sh.workgroups) return hipErrorInvalidValue;
    const bp_lds_fn k = hybrid ? bp_lds_hybrid_kernel() : bp_lds_kernel_for(dc, dv, sh.threads != BPL_THREADS);
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.lds);
    if (e != hipSuccess) return e;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(a.n_shots, sh.workgroups);
    if (switches().debug)
        std::fprintf(stderr, "[ldpc_hip] bp lds form: %zu B of LDS per shot (graph copies: %u), %u workgroups of %u threads\n", sh.lds, sh.copies, grid, sh.threads);
    g_kernel_name = hybrid ? "bp_lds_kernel<6,3> (hybrid: messages in LDS and L2)" : dc == 4 ? "bp_lds_kernel<4>" : dc == 6 ? "bp_lds_kernel<6>" : dc == 7 ? "bp_lds_kernel<7>" : dc == 3 ? "bp_lds_kernel<3>" : "bp_lds_kernel<0>";
    hipLaunchKernelGGL(k, dim3(grid), dim3(sh.threads), sh.lds, st, g, a, sh.copies, hybrid ? bp_lds_hybrid_edges(g) : g.E);
    return hipGetLastError();
}

size_t bp_stream_state_bytes(uint32_t groups) { return (size_t)groups * (5 * 8 + 64 * (4 + 8 + 4 + 4) + 4) + 64; }

// Runs the streamed BP to completion on `st`.  `state` = bp_stream_state_bytes(groups) bytes of device memory.
BpsState bp_stream_state_at(void *state, uint32_t groups)
{
    BpsState s;
    char *p = (char *)state;
    s.alive = (unsigned long long *)p; p += 64;
    s.act = (uint64_t *)p; p += (size_t)groups * 8;
    s.fresh = (uint64_t *)p; p += (size_t)groups * 8;
    s.take = (uint64_t *)p; p += (size_t)groups * 8;
    s.mism = (uint64_t *)p; p += (size_t)groups * 8;
    s.myi = (uint64_t *)p; p += (size_t)groups * 64 * 8;
    s.iter = (uint32_t *)p; p += (size_t)groups * 64 * 4;
    s.flags = (uint32_t *)p; p += (size_t)groups * 64 * 4;
    s.gmap_store = (uint32_t *)p; p += (size_t)groups * 4;
    s.moves = (uint32_t *)p;
    s.gmap = nullptr;
    return s;
}

hipError_t bp_stream_run(const DeviceGraph &g, const BpArgs &a, void *state, uint32_t groups, int dc, int dv, hipStream_t st,
                         uint64_t *rounds_out, bool resume)
{
    if (!a.n_shots || !groups) return hipSuccess;
    hipError_t e = hipSuccess;
    if (!resume && (e = hipMemsetAsync(state, 0, bp_stream_state_bytes(groups), st)) != hipSuccess) return e;
    BpsState s = bp_stream_state_at(state, groups);
    // parts: enough blocks to fill the chip a few times over, no more than the work items of the smallest phase
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t Wn = (g.n + 63) / 64;
    // The bit update works a 64-bit chunk per wave, or half / a quarter of one when whole chunks would leave the
    // chip's 16 waves per CU mostly empty (small codes, the few groups of a tail): a piece's edge rows are fetched
    // one batch after the other, so the phase lasts as long as its longest piece however idle the chip is.
    uint32_t parts = 1, bsplit = 1;
    auto retile = [&](uint32_t in_play) {
        const uint64_t chunks = (uint64_t)in_play * Wn, room = 16ull * (uint64_t)cus;
        bsplit = chunks >= room ? 1u : 2 * chunks >= room ? 2u : 4u;
        parts = std::max<uint32_t>(1, (uint32_t)(8 * cus) / in_play);
        parts = std::min<uint32_t>(parts, std::max<uint32_t>(1, (Wn * bsplit + BPS_WAVES - 1) / BPS_WAVES));
    };
    retile(groups);
    const dim3 blk(BPS_THREADS), one(64);
    // two sets of groups half an iteration apart (see bps_iterate); a single group runs alone
    // Measured at C5: no gain from the skew (a check-update wave keeps its SIMD only 22 % busy, so the check update
    // needs all four wave slots of a SIMD to itself); one set by default, LDPC_HIP_BP_STREAM_SKEW=1 for experiments
    const bool skew = !resume && switches().bp_stream_skew;
    if (!resume) g_kernel_name = "bps_iterate (streamed form)";
    uint32_t set0[2] = {0, skew ? (groups + 1) / 2 : groups}, setn[2] = {skew ? (groups + 1) / 2 : groups, skew ? groups / 2 : 0};
    unsigned long long alive[2] = {0, 0};
    bool wants_checks[2] = {true, true};
    bool tail = false;   // resume mode after the first round: nothing can be refilled, short control
    // stop test, harvest, refill, set-up, stop test and harvest of the new shots (a zero syndrome stops at once)
    // In the tail the host reads the alive count (one sync) only every `burst` iterations: rounds on shots that
    // have stopped meanwhile do nothing (stopped slots are harvested once and leave the active mask; groups
    // without active slots return at once), and a round of a small tail costs less than the sync.
    uint32_t burst = 4, since_sync = 0;
    if (switches().bp_tail_burst) burst = (uint32_t)switches().bp_tail_burst;   // A/B aid
    auto control = [&](int t) -> hipError_t {
        const dim3 grid(setn[t] * parts), ctl(setn[t]);
        const uint32_t g0 = set0[t];
        hipError_t err;
        if (tail) {
            hipLaunchKernelGGL(bps_stop, grid, blk, 0, st, g, a, s, parts, g0, 0);
            hipLaunchKernelGGL(bps_harvest_words, grid, blk, 0, st, g, a, s, parts, g0, 0);
            if ((err = hipMemsetAsync(s.alive + t, 0, 8, st)) != hipSuccess) return err;
            hipLaunchKernelGGL(bps_ctl_harvest, ctl, one, 0, st, a, s, g0, 0, s.alive + t);
        } else {
            hipLaunchKernelGGL(bps_stop, grid, blk, 0, st, g, a, s, parts, g0, 0);
            hipLaunchKernelGGL(bps_harvest_words, grid, blk, 0, st, g, a, s, parts, g0, 0);
            hipLaunchKernelGGL(bps_ctl_harvest, ctl, one, 0, st, a, s, g0, 0, (unsigned long long *)nullptr);
            hipLaunchKernelGGL(bps_ctl_refill, ctl, one, 0, st, a, s, g0);
            hipLaunchKernelGGL(bps_setup_words, grid, blk, 0, st, g, a, s, parts, g0);
            hipLaunchKernelGGL(bps_setup_synd, grid, blk, 0, st, g, a, s, parts, g0);
            hipLaunchKernelGGL(bps_stop, grid, blk, 0, st, g, a, s, parts, g0, 1);
            hipLaunchKernelGGL(bps_harvest_words, grid, blk, 0, st, g, a, s, parts, g0, 1);
            if ((err = hipMemsetAsync(s.alive + t, 0, 8, st)) != hipSuccess) return err;
            hipLaunchKernelGGL(bps_ctl_harvest, ctl, one, 0, st, a, s, g0, 1, s.alive + t);
        }
        if (tail && ++since_sync < burst) return hipSuccess;   // alive[t] stays as last read (non-zero)
        since_sync = 0;
        if ((err = hipMemcpyAsync(&alive[t], s.alive + t, 8, hipMemcpyDeviceToHost, st)) != hipSuccess) return err;
        return hipStreamSynchronize(st);
    };
    auto iterate = [&](int c, int b) {
        const uint32_t gc0 = c >= 0 ? set0[c] : 0, gcn = c >= 0 ? setn[c] : 0, gb0 = b >= 0 ? set0[b] : 0, gbn = b >= 0 ? setn[b] : 0;
        const dim3 grid((gcn + gbn) * parts);
        const bool small_dv = dv == -4 || (dv >= 1 && dv <= 4);   // as in bp_kernel_for
        if (dc == 6 && dv == 3) hipLaunchKernelGGL((bps_iterate<6, 3>), grid, blk, 0, st, g, a, s, parts, gc0, gcn, gb0, gbn, bsplit);
        else if (dc == 4 && dv == 3) hipLaunchKernelGGL((bps_iterate<4, 3>), grid, blk, 0, st, g, a, s, parts, gc0, gcn, gb0, gbn, bsplit);
        else if (dc == 7 && small_dv) hipLaunchKernelGGL((bps_iterate<7, -4>), grid, blk, 0, st, g, a, s, parts, gc0, gcn, gb0, gbn, bsplit);
        else if (dc == 7) hipLaunchKernelGGL((bps_iterate<7, 0>), grid, blk, 0, st, g, a, s, parts, gc0, gcn, gb0, gbn, bsplit);
        else if (small_dv) hipLaunchKernelGGL((bps_iterate<0, -4>), grid, blk, 0, st, g, a, s, parts, gc0, gcn, gb0, gbn, bsplit);
        else hipLaunchKernelGGL((bps_iterate<0, 0>), grid, blk, 0, st, g, a, s, parts, gc0, gcn, gb0, gbn, bsplit);
    };
    // Tail compaction (bps_compact): when the shots left fill at most 3/4 of the slots of the groups in play, move
    // them into the first ceil(alive / 64) of those groups and drop the rest.  Called after a control (the stream is
    // idle, the state at an iteration boundary); a round costs a pass over every group in play, a compaction less
    // than one such pass.
    std::vector<uint32_t> glist;   // host copy of the group list behind s.gmap (empty: identity)
    uint64_t compactions = 0;
    auto compact = [&]() -> hipError_t {
        const uint32_t cur = setn[0];
        hipError_t err;
        std::vector<uint64_t> act(groups), fr(groups);
        if ((err = hipMemcpyAsync(act.data(), s.act, (size_t)groups * 8, hipMemcpyDeviceToHost, st)) != hipSuccess) return err;
        if ((err = hipMemcpyAsync(fr.data(), s.fresh, (size_t)groups * 8, hipMemcpyDeviceToHost, st)) != hipSuccess) return err;
        if ((err = hipStreamSynchronize(st)) != hipSuccess) return err;
        auto phys = [&](uint32_t i) { return glist.empty() ? i : glist[i]; };
        uint64_t total = 0;
        for (uint32_t i = 0; i < cur; i++) {
            if (fr[phys(i)] & act[phys(i)]) return hipSuccess;   // a fresh slot has no messages yet: wait a round
            total += (uint64_t)__builtin_popcountll(act[phys(i)]);
        }
        const uint32_t keep = (uint32_t)((total + 63) / 64);
        if (keep == 0 || keep >= cur) return hipSuccess;
        std::vector<uint32_t> src, mv((size_t)keep * 64, 0xFFFFFFFFu);
        for (uint32_t i = keep; i < cur; i++) {
            for (uint64_t mleft = act[phys(i)]; mleft; mleft &= mleft - 1)
                src.push_back(phys(i) * 64u + (uint32_t)__builtin_ctzll(mleft));
            act[phys(i)] = 0;
        }
        for (uint32_t jd = 0; jd < keep && !src.empty(); jd++)
            for (uint32_t slot = 0; slot < 64 && !src.empty(); slot++)
                if (!((act[phys(jd)] >> slot) & 1ull)) {
                    mv[(size_t)jd * 64 + slot] = src.back();
                    src.pop_back();
                    act[phys(jd)] |= 1ull << slot;
                    fr[phys(jd)] &= ~(1ull << slot);   // (a shot harvested in its first round leaves the bit set)
                }
        if (!src.empty()) return hipErrorLaunchFailure;   // cannot happen: keep * 64 >= total
        if ((err = hipMemcpyAsync(s.moves, mv.data(), mv.size() * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return err;
        uint32_t cparts = std::max<uint32_t>(1, (uint32_t)(8 * cus) / keep);
        cparts = std::min<uint32_t>(cparts, std::max<uint32_t>(1, (g.E + BPS_WAVES - 1) / BPS_WAVES));
        hipLaunchKernelGGL(bps_compact, dim3(keep * cparts), blk, 0, st, g, a, s, cparts, (const uint32_t *)s.moves, (a.post_out || a.chk_out) ? 1 : 0);
        if ((err = hipMemcpyAsync(s.act, act.data(), (size_t)groups * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return err;
        if ((err = hipMemcpyAsync(s.fresh, fr.data(), (size_t)groups * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return err;
        if ((err = hipStreamSynchronize(st)) != hipSuccess) return err;   // (the host vectors go out of scope)
        setn[0] = keep;
        retile(keep);
        compactions++;
        return hipGetLastError();
    };
    const bool may_compact = !skew && !switches().bp_no_compact;   // (A/B and test aid)
    if (resume) hipLaunchKernelGGL(bps_parity_init, dim3(groups * parts), blk, 0, st, g, a, s, parts, 0u);
    for (int t = 0; t < 2; t++)
        if (setn[t] && (e = control(t)) != hipSuccess) return e;
    if (resume && alive[0]) {
        // tail of the persistent kernel: few groups still hold shots.  List them once (the list only shrinks:
        // groups that finish later return at once), give each many more parts, and drop refill / set-up from
        // the round
        std::vector<uint64_t> act(groups);
        if ((e = hipMemcpy(act.data(), s.act, (size_t)groups * 8, hipMemcpyDeviceToHost)) != hipSuccess) return e;
        std::vector<uint32_t> list;
        for (uint32_t i = 0; i < groups; i++)
            if (act[i]) list.push_back(i);
        if ((e = hipMemcpy(s.gmap_store, list.data(), list.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return e;
        s.gmap = s.gmap_store;
        setn[0] = (uint32_t)list.size();
        retile(setn[0]);
        glist = list;
        tail = true;
        if (may_compact && setn[0] > 1 && alive[0] <= 48ull * setn[0] && (e = compact()) != hipSuccess) return e;
    }
    // a call whose shots were all loaded by the first control has nothing left to refill either: short control
    if (!resume && a.n_shots <= 64ull * groups) tail = true;
    const uint64_t max_rounds = 2 * burst + 2 * (((a.n_shots + 64ull * groups - 1) / (64ull * groups) + 1) * ((uint64_t)a.max_iter + 2) + 2);
    uint64_t rounds = 0;
    if (alive[0] && alive[1]) {   // skew the sets: set 0 starts half an iteration early
        iterate(0, -1);
        wants_checks[0] = false;
    }
    while (alive[0] || alive[1]) {
        if (++rounds > max_rounds) return hipErrorLaunchFailure;   // cannot happen: every round stops or advances every active slot
        int c = -1, b = -1;
        for (int t = 1; t >= 0; t--)
            if (alive[t]) {
                if (wants_checks[t]) c = t;
                else b = t;
            }
        iterate(c, b);
        if (c >= 0) wants_checks[c] = false;
        if (b >= 0) {
            hipLaunchKernelGGL(bps_ctl_advance, dim3(setn[b]), one, 0, st, s, set0[b]);
            if ((e = control(b)) != hipSuccess) return e;
            wants_checks[b] = true;
            if (tail && since_sync == 0 && may_compact && setn[0] > 1 && alive[0] && alive[0] <= 48ull * setn[0] &&
                (e = compact()) != hipSuccess)
                return e;
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (rounds_out) *rounds_out = rounds;
    if (switches().debug && compactions) std::fprintf(stderr, "[ldpc_hip] bp stream: %llu tail compactions\n", (unsigned long long)compactions);
    return hipGetLastError();
}

hipError_t launch_bp_math_probe(int kind, uint64_t n, const double *d_in, double *d_out, hipStream_t st)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(bp_math_probe_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, st, kind, n, d_in, d_out);
    return hipGetLastError();
}

hipError_t launch_count_flags(const uint8_t *d_flags, const uint8_t *d_flags2, uint64_t n, unsigned long long *d_counters,
                              hipStream_t st)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(count_flags_kernel, dim3(grid_for(n, 256 * 16, 2048)), dim3(256), 0, st, d_flags, d_flags2, n, d_counters);
    return hipGetLastError();
}

}  // namespace ldpc
