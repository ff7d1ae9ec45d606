// Shared device-side definitions: graph layout in HBM, Philox4x32-10 and the
// noise specification (DESIGN.md "Noise specification").
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ldpc {

// Tanner graph in HBM, CSR both ways, u32 indices.  Edge order is the
// reference's Edges order (edges.rs:23-29): check-major, bits ascending.
struct DeviceGraph {
    uint32_t n;             // bits
    uint32_t m;             // checks
    uint32_t E;             // edges
    uint32_t max_dv;        // largest bit degree
    uint32_t max_dc;        // largest check degree
    const uint32_t *h_ptr;  // [m+1]  check -> first edge
    const uint32_t *h_idx;  // [E]    edge  -> bit
    const uint32_t *t_ptr;  // [n+1]  bit   -> first slot of bit_adjacencies
    const uint32_t *t_idx;  // [E]    slot  -> check (ascending inside a bit)
    const uint32_t *t_edge; // [E]    slot  -> edge index in check-major order
    const uint32_t *t_pack; // [E]    slot  -> edge | check << 16 (one load for both; null unless E and m are below 65536)
    const uint4 *t_pack4;   // [n]    bit   -> its (at most four) t_pack words, unused ones 0xFFFFFFFF (null unless t_pack exists and max_dv <= 4)
};

// Geometric gap table of a channel: P(gap >= j) = q[j] / 2^32, j = 1..64.
struct GapTable {
    uint32_t q[65];
    uint32_t p_is_zero;
    float inv_log2q;   // 1 / log2(1 - p) (0 when p == 1): starting point of gap_of's table search
};

__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#ifdef __HIPCC__
// Draw stream of one (shot, word): Philox counter (shot_lo, shot_hi, word, block),
// key = seed; 32-bit draws are consumed in order x, y, z, w of successive blocks.
struct DrawStream {
    uint32_t s_lo, s_hi, word, block, k0, k1;
    uint32_t buf[4];
    int have;
    __device__ void init(uint64_t seed, uint64_t shot, uint32_t w)
    {
        k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
        s_lo = (uint32_t)shot; s_hi = (uint32_t)(shot >> 32);
        word = w; block = 0; have = 0;
    }
    __device__ uint32_t draw()
    {
        if (have == 0) {
            philox4x32_10(s_lo, s_hi, word, block, k0, k1, buf);
            block++;
            have = 4;
        }
        uint32_t i = 4 - have;
        have--;
        // select without dynamic register indexing
        return i == 0 ? buf[0] : (i == 1 ? buf[1] : (i == 2 ? buf[2] : buf[3]));
    }
};

// gap(u) = #{ j in 1..64 : u < q[j] }, q non-increasing.  `q` may live in LDS or global memory.
// The count is the largest j with u < q[j]; q[j] ~ (1-p)^j 2^32, so j ~ (log2(u) - 32) / log2(1-p): a float
// estimate picks the starting index and the table decides (the result is exact whatever the estimate; a good
// one costs one round trip to the table instead of the six of a binary search).
__device__ inline uint32_t gap_of(const uint32_t *q, float inv_log2q, uint32_t u)
{
    const float est = (__log2f((float)u) - 32.0f) * inv_log2q;       // u = 0: +inf; p = 1: NaN -> 0 below
    uint32_t j = (uint32_t)fminf(fmaxf(est, 0.0f), 64.0f);
    const uint32_t a = q[j], b = q[j < 64 ? j + 1 : 64];
    if (j < 64 && u < b) {
        do j++;
        while (j < 64 && u < q[j + 1]);
    } else if (j > 0 && !(u < a)) {
        do j--;
        while (j > 0 && !(u < q[j]));
    }
    return j;
}

// One 64-bit word of a BSC sample (binary_symmetric_channel.rs:35-43 distribution).  The draws of a Philox block are
// consumed in order x, y, z, w exactly as the sequential definition (DESIGN.md "Noise specification") says; the four
// gaps of a block are looked up side by side first (independent table searches overlap), then the positions follow in
// a short chain of adds.  Draws of the last block that the word does not need are discarded, as before.
__device__ inline uint64_t bsc_word(const uint32_t *q, float inv_log2q, bool p_is_zero, uint64_t seed, uint64_t shot, uint32_t w,
                                    uint32_t nbits)
{
    if (p_is_zero) return 0;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), s_lo = (uint32_t)shot, s_hi = (uint32_t)(shot >> 32);
    uint64_t word = 0;
    uint32_t pos = 0;
    for (uint32_t block = 0;; block++) {
        uint32_t u[4], g[4];
        philox4x32_10(s_lo, s_hi, w, block, k0, k1, u);
#pragma unroll
        for (int k = 0; k < 4; k++) g[k] = gap_of(q, inv_log2q, u[k]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            pos += g[k];
            if (pos >= nbits) return word;
            word |= 1ull << pos;
            pos++;
            if (pos >= nbits) return word;
        }
    }
}

// One Philox block of a word's BSC stream, for callers that keep (word, pos) themselves: the same draws, gaps and
// positions as bsc_word's loop body, with the usual case of a gap look-up (the float estimate is the answer: q[j] > u
// >= q[j + 1]) checked for the four draws together and without branches.  `q` must hold q[65] = 0 after the table.
// Returns true when the word is complete.
__device__ inline bool bsc_block(const uint32_t *q, float inv_log2q, uint32_t k0, uint32_t k1, uint32_t s_lo, uint32_t s_hi, uint32_t w,
                                 uint32_t block, uint32_t nbits, uint64_t &word, uint32_t &pos)
{
    uint32_t u[4], g[4];
    philox4x32_10(s_lo, s_hi, w, block, k0, k1, u);
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float est = (__log2f((float)u[k]) - 32.0f) * inv_log2q;
        const uint32_t j = (uint32_t)fminf(fmaxf(est, 0.0f), 64.0f);
        const uint32_t a = q[j], b = q[j + 1];
        // (q[0] = 2^32 - 1 stands for 2^32: the one draw it misjudges, u = 2^32 - 1 at j = 0, only takes the exact path)
        bad |= (u[k] < b) | !(u[k] < a);
        g[k] = j;
    }
    if (bad) {
#pragma unroll
        for (int k = 0; k < 4; k++) g[k] = gap_of(q, inv_log2q, u[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        pos += g[k];
        word |= pos < nbits ? 1ull << (pos & 63) : 0ull;
        pos++;
    }
    return pos >= nbits;   // (pos only grows: once past the end nothing more is set)
}

// One word of a depolarizing sample (depolarizing.rs:34-48): x part = X or Y,
// z part = Z or Y (css.rs:92-99).
__device__ inline void depol_word(const uint32_t *q, float inv_log2q, bool p_is_zero, uint64_t seed, uint64_t shot, uint32_t w,
                                  uint32_t nbits, uint64_t &xw, uint64_t &zw)
{
    xw = 0; zw = 0;
    if (p_is_zero) return;
    DrawStream s;
    s.init(seed, shot, w);
    uint32_t pos = 0;
    for (;;) {
        pos += gap_of(q, inv_log2q, s.draw());
        if (pos >= nbits) break;
        uint32_t pauli = (uint32_t)(((uint64_t)s.draw() * 3u) >> 32);   // 0 = X, 1 = Y, 2 = Z
        if (pauli <= 1) xw |= 1ull << pos;
        if (pauli >= 1) zw |= 1ull << pos;
        pos++;
        if (pos >= nbits) break;
    }
}

__device__ inline uint32_t bits_in_word(uint32_t n, uint32_t w)
{
    uint32_t rem = n - w * 64u;
    return rem < 64u ? rem : 64u;
}
#endif  // __HIPCC__

}  // namespace ldpc
