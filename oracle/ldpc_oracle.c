/*
 * ldpc_oracle.c -- CPU restatement of the Monte-Carlo decoding path of the
 * `ldpc` Rust crate (maxtremblay/ldpc 0.7.0).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
 * in __graft_entry__.py and the `cpu_baseline` leg of bench.py may load it.
 * The product path (ldpc_amd/csrc) never links, loads or calls anything here.
 *
 * Parity status: the Rust reference cannot be built in this environment (no
 * rustc/cargo, dependencies not vendored), so this restatement is pinned by
 * the reference's own known-answer tests (tests/golden/reference_kats.json,
 * see tests/test_oracle_kats.py):
 *   flip.rs:60-98, belief_propagation.rs:190-230, linear_code/mod.rs:376-388,
 *   421-425, edges.rs:36-56, css/mod.rs:98-111, 187-199, 213-224, 236-246.
 * PARITY UNPINNED (the reference holds no fixture): the random streams of the
 * `rand` crate (noise samples) and the graphs drawn by `bigs`.  The noise
 * sampler below is therefore this project's own specification (Philox4x32-10,
 * geometric gap sampling); it reproduces the reference's *distribution*
 * (binary_symmetric_channel.rs:35-43, depolarizing.rs:34-48), not its stream.
 *
 * Every function cites the reference lines it follows.
 *
 * Build: gcc -O2 -fno-fast-math -ffp-contract=off -fopenmp -shared -fPIC
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* Sparse binary matrix: rows of strictly ascending column indices.          */
/* Stands for sparse-bin-mat's SparseBinMat (third-party, not vendored;      */
/* semantics per SURVEY.md appendix A).                                      */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint32_t n;        /* number of columns (bits)            */
    uint32_t m;        /* number of rows (checks)             */
    int64_t E;         /* number of ones (edges)              */
    int64_t *h_ptr;    /* [m+1] row starts                    */
    uint32_t *h_idx;   /* [E] columns, ascending inside a row */
    int64_t *t_ptr;    /* [n+1] transposed (bit_adjacencies)  */
    uint32_t *t_idx;   /* [E] checks, ascending inside a bit  */
    int64_t *t_edge;   /* [E] CSC position -> CSR edge index  */
    int64_t *h_csc;    /* [E] CSR edge index -> CSC position  */
} orc_code;

void orc_code_free(orc_code *c)
{
    if (!c) return;
    free(c->h_ptr); free(c->h_idx); free(c->t_ptr); free(c->t_idx);
    free(c->t_edge); free(c->h_csc); free(c);
}

/* SparseBinMat::new(n, rows) + transposed() (linear_code/mod.rs:90-98).
 * Returns NULL when a row is not strictly ascending or out of range (the
 * reference panics in that case). */
orc_code *orc_code_new(uint32_t n, uint32_t m, const int64_t *ptr, const uint32_t *idx)
{
    if (ptr[0] != 0) return NULL;
    for (uint32_t r = 0; r < m; r++) {
        if (ptr[r + 1] < ptr[r]) return NULL;
        for (int64_t e = ptr[r]; e < ptr[r + 1]; e++) {
            if (idx[e] >= n) return NULL;
            if (e > ptr[r] && idx[e] <= idx[e - 1]) return NULL;
        }
    }
    orc_code *c = (orc_code *)calloc(1, sizeof(orc_code));
    int64_t E = ptr[m];
    c->n = n; c->m = m; c->E = E;
    c->h_ptr = (int64_t *)malloc(sizeof(int64_t) * ((size_t)m + 1));
    c->h_idx = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(E ? E : 1));
    c->t_ptr = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    c->t_idx = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(E ? E : 1));
    c->t_edge = (int64_t *)malloc(sizeof(int64_t) * (size_t)(E ? E : 1));
    c->h_csc = (int64_t *)malloc(sizeof(int64_t) * (size_t)(E ? E : 1));
    memcpy(c->h_ptr, ptr, sizeof(int64_t) * ((size_t)m + 1));
    if (E) memcpy(c->h_idx, idx, sizeof(uint32_t) * (size_t)E);
    for (int64_t e = 0; e < E; e++) c->t_ptr[idx[e] + 1]++;
    for (uint32_t b = 0; b < n; b++) c->t_ptr[b + 1] += c->t_ptr[b];
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
    memcpy(fill, c->t_ptr, sizeof(int64_t) * ((size_t)n + 1));
    /* rows visited in ascending order => checks ascending inside each bit */
    for (uint32_t r = 0; r < m; r++)
        for (int64_t e = ptr[r]; e < ptr[r + 1]; e++) {
            int64_t k = fill[idx[e]]++;
            c->t_idx[k] = r;
            c->t_edge[k] = e;
            c->h_csc[e] = k;
        }
    free(fill);
    return c;
}

uint32_t orc_code_len(const orc_code *c) { return c->n; }
uint32_t orc_code_num_checks(const orc_code *c) { return c->m; }
int64_t orc_code_num_edges(const orc_code *c) { return c->E; }

/* Edges iterator order (edges.rs:23-29): row-major (check, bit). */
void orc_code_edges(const orc_code *c, uint32_t *bits, uint32_t *checks)
{
    for (uint32_t r = 0; r < c->m; r++)
        for (int64_t e = c->h_ptr[r]; e < c->h_ptr[r + 1]; e++) {
            bits[e] = c->h_idx[e];
            checks[e] = r;
        }
}

void orc_code_bit_adjacencies(const orc_code *c, int64_t *ptr, uint32_t *idx)
{
    memcpy(ptr, c->t_ptr, sizeof(int64_t) * ((size_t)c->n + 1));
    if (c->E) memcpy(idx, c->t_idx, sizeof(uint32_t) * (size_t)c->E);
}

/* ------------------------------------------------------------------------ */
/* Sorted position lists: stand for SparseBinVec.                            */
/* ------------------------------------------------------------------------ */

/* `&H * &v` (linear_code/mod.rs:404): out position r set iff |row r ∩ v| odd.
 * Sorted-list intersection per row, as a sparse library would do it. */
static size_t spmv_pos(const orc_code *c, const uint32_t *pos, size_t npos, uint32_t *out)
{
    size_t nout = 0;
    for (uint32_t r = 0; r < c->m; r++) {
        int64_t e = c->h_ptr[r], e1 = c->h_ptr[r + 1];
        size_t k = 0;
        unsigned par = 0;
        while (e < e1 && k < npos) {
            uint32_t a = c->h_idx[e], b = pos[k];
            if (a == b) { par ^= 1; e++; k++; }
            else if (a < b) e++;
            else k++;
        }
        if (par) out[nout++] = r;
    }
    return nout;
}

/* LinearCode::syndrome_of (linear_code/mod.rs:393-405).  Returns -1 on length
 * mismatch (the reference panics). */
int64_t orc_syndrome_pos(const orc_code *c, uint32_t len, const uint32_t *pos, size_t npos,
                         uint32_t *out)
{
    if (len != c->n) return -1;
    return (int64_t)spmv_pos(c, pos, npos, out);
}

/* `&a + &b` on SparseBinVec: symmetric difference of sorted lists. */
static size_t xor_pos(const uint32_t *a, size_t na, const uint32_t *b, size_t nb, uint32_t *out)
{
    size_t i = 0, j = 0, k = 0;
    while (i < na && j < nb) {
        if (a[i] == b[j]) { i++; j++; }
        else if (a[i] < b[j]) out[k++] = a[i++];
        else out[k++] = b[j++];
    }
    while (i < na) out[k++] = a[i++];
    while (j < nb) out[k++] = b[j++];
    return k;
}

/* SparseBinVec::is_one_at: binary search in the sorted list. */
static int is_one_at(const uint32_t *pos, size_t npos, uint32_t x)
{
    size_t lo = 0, hi = npos;
    while (lo < hi) {
        size_t mid = (lo + hi) >> 1;
        if (pos[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo < npos && pos[lo] == x;
}

/* ------------------------------------------------------------------------ */
/* FlipDecoder (flip.rs:21-43), restated as written: rescan the rows of      */
/* bit_adjacencies from bit 0 after every flip, full H * e_bit per flip.      */
/* ------------------------------------------------------------------------ */

/* find_flippable (flip.rs:35-43) */
static int64_t find_flippable(const orc_code *c, const uint32_t *syn, size_t nsyn)
{
    for (uint32_t b = 0; b < c->n; b++) {
        int64_t k0 = c->t_ptr[b], k1 = c->t_ptr[b + 1];
        int64_t unsat = 0;
        for (int64_t k = k0; k < k1; k++) unsat += is_one_at(syn, nsyn, c->t_idx[k]);
        if (unsat > (k1 - k0) / 2) return b;
    }
    return -1;
}

/* FlipDecoder::decode (flip.rs:21-33).  in/out are sorted positions; out has
 * room for n entries.  Returns the output weight, -1 on length mismatch. */
int64_t orc_flip_decode_pos(const orc_code *c, uint32_t len, const uint32_t *pos, size_t npos,
                            uint32_t *out, uint64_t *nflips)
{
    if (len != c->n) return -1;
    uint32_t *syn = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)c->m + 1));
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)c->m + 1));
    uint32_t *upd = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)c->m + 1));
    uint32_t *o2 = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)c->n + 1));
    size_t nsyn = spmv_pos(c, pos, npos, syn);              /* :25 */
    size_t nout = npos;
    memcpy(out, pos, sizeof(uint32_t) * npos);              /* :26 */
    uint64_t flips = 0;
    for (;;) {
        int64_t b = find_flippable(c, syn, nsyn);           /* :27 */
        if (b < 0) break;
        uint32_t bit = (uint32_t)b;
        size_t nupd = spmv_pos(c, &bit, 1, upd);            /* :28-29 syndrome_of(update) */
        size_t nt = xor_pos(syn, nsyn, upd, nupd, tmp);
        memcpy(syn, tmp, sizeof(uint32_t) * nt); nsyn = nt;
        size_t no = xor_pos(out, nout, &bit, 1, o2);        /* :30 */
        memcpy(out, o2, sizeof(uint32_t) * no); nout = no;
        flips++;
    }
    if (nflips) *nflips = flips;
    free(syn); free(tmp); free(upd); free(o2);
    return (int64_t)nout;
}

/* ------------------------------------------------------------------------ */
/* Bit-packed helpers (u64 words, bit i of a vector = word i/64, bit i%64).  */
/* ------------------------------------------------------------------------ */
static inline int getbit(const uint64_t *w, uint32_t i) { return (int)((w[i >> 6] >> (i & 63)) & 1u); }
static inline void flipbit(uint64_t *w, uint32_t i) { w[i >> 6] ^= (uint64_t)1 << (i & 63); }
static inline size_t nwords(uint32_t nbits) { return ((size_t)nbits + 63) >> 6; }

size_t orc_words_to_pos(const uint64_t *w, uint32_t nbits, uint32_t *pos)
{
    size_t k = 0;
    for (uint32_t i = 0; i < nbits; i++) if (getbit(w, i)) pos[k++] = i;
    return k;
}

void orc_pos_to_words(const uint32_t *pos, size_t npos, uint32_t nbits, uint64_t *w)
{
    memset(w, 0, nwords(nbits) * 8);
    for (size_t k = 0; k < npos; k++) flipbit(w, pos[k]);
}

/* s = H e over GF(2), packed (same arithmetic as spmv_pos). */
void orc_syndrome_words(const orc_code *c, const uint64_t *e, uint64_t *s)
{
    memset(s, 0, nwords(c->m) * 8);
    for (uint32_t r = 0; r < c->m; r++) {
        unsigned par = 0;
        for (int64_t k = c->h_ptr[r]; k < c->h_ptr[r + 1]; k++) par ^= (unsigned)getbit(e, c->h_idx[k]);
        if (par) flipbit(s, r);
    }
}

/* Packed, incremental flip decoder.  Same result as orc_flip_decode_pos (the
 * decision depends only on the current syndrome: SURVEY.md 3.3) but O(deg)
 * per flip; tests cross-check the two and use this one for large sweeps. */
void orc_flip_decode_words(const orc_code *c, const uint64_t *in, uint64_t *out, uint64_t *syn_out,
                           uint64_t *nflips)
{
    size_t wn = nwords(c->n), wm = nwords(c->m);
    uint64_t *s = (uint64_t *)malloc((wm ? wm : 1) * 8);
    uint8_t *unsat = (uint8_t *)calloc((size_t)c->n + 1, 1);
    memcpy(out, in, wn * 8);
    orc_syndrome_words(c, in, s);
    for (uint32_t b = 0; b < c->n; b++)
        for (int64_t k = c->t_ptr[b]; k < c->t_ptr[b + 1]; k++) unsat[b] += (uint8_t)getbit(s, c->t_idx[k]);
    uint64_t flips = 0;
    uint32_t b = 0;
    while (b < c->n) {
        int64_t deg = c->t_ptr[b + 1] - c->t_ptr[b];
        if ((int64_t)unsat[b] > deg / 2) {
            uint32_t lowest = b + 1;
            flipbit(out, b);
            for (int64_t k = c->t_ptr[b]; k < c->t_ptr[b + 1]; k++) {
                uint32_t chk = c->t_idx[k];
                flipbit(s, chk);
                int now = getbit(s, chk);
                for (int64_t e = c->h_ptr[chk]; e < c->h_ptr[chk + 1]; e++) {
                    uint32_t bb = c->h_idx[e];
                    unsat[bb] = (uint8_t)(unsat[bb] + (now ? 1 : -1));
                    if (bb < lowest) lowest = bb;   /* an earlier bit may have become flippable */
                }
            }
            flips++;
            b = lowest;   /* restart from the smallest index whose state changed */
        } else {
            b++;
        }
    }
    if (syn_out) memcpy(syn_out, s, wm * 8);
    if (nflips) *nflips = flips;
    free(s); free(unsat);
}

/* ------------------------------------------------------------------------ */
/* BpDecoder (belief_propagation.rs).                                        */
/* ------------------------------------------------------------------------ */

/* f64::atanh of Rust's std (library/std/src/f64.rs; std is not vendored and
 * its version is unpinned): 0.5 * ln_1p(2x / (1 - x)). */
static double rust_atanh(double x) { return 0.5 * log1p((2.0 * x) / (1.0 - x)); }

/* SyndromeDecoder::correction_for (belief_propagation.rs:23-30) with
 * initialize_from (:48-76), BpState::decode (:100-114), update_once (:133-140),
 * update_checks (:150-166), update_bits (:168-180).
 *  syn:   packed syndrome (m bits)
 *  corr:  packed correction out (n bits)
 *  iters: iterations used
 *  post:  optional [n] posterior LLRs at stop
 *  chk:   optional [E] check->bit messages at stop, CSR edge order
 */
void orc_bp_correction_for(const orc_code *c, double p, uint32_t max_iter, const uint64_t *syn,
                           uint64_t *corr, uint32_t *iters, double *post, double *chk)
{
    int64_t E = c->E;
    uint32_t n = c->n, m = c->m;
    double prior = log((1.0 - p) / p);                                   /* :38 */
    double *bits = (double *)malloc(sizeof(double) * (size_t)(E ? E : 1));   /* CSR  :60-66 */
    double *checks = (double *)malloc(sizeof(double) * (size_t)(E ? E : 1)); /* CSC  :68-76 */
    double *products = (double *)malloc(sizeof(double) * ((size_t)m + 1));
    double *sums = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    double *lik = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    uint64_t *dec = (uint64_t *)malloc((nwords(n) ? nwords(n) : 1) * 8);
    for (int64_t e = 0; e < E; e++) { bits[e] = prior; checks[e] = 0.0; }
    uint32_t it = 0;
    for (;;) {
        /* BpState::decode :100-114 : posterior starts from the prior, adds the
         * check messages of the bit in ascending check order */
        memset(dec, 0, nwords(n) * 8);
        for (uint32_t b = 0; b < n; b++) {
            double l = prior;
            for (int64_t k = c->t_ptr[b]; k < c->t_ptr[b + 1]; k++) l += checks[k];
            lik[b] = l;
            if (l < 0.0) flipbit(dec, b);
        }
        /* stop test :26-27 : H * decode == syndrome, or iteration cap */
        int same = 1;
        for (uint32_t r = 0; r < m && same; r++) {
            unsigned par = 0;
            for (int64_t e = c->h_ptr[r]; e < c->h_ptr[r + 1]; e++) par ^= (unsigned)getbit(dec, c->h_idx[e]);
            if ((int)par != getbit(syn, r)) same = 0;
        }
        if (same || it == max_iter) break;
        it++;                                                            /* :134 */
        /* update_checks :150-166 */
        for (uint32_t r = 0; r < m; r++) {
            double prod = 1.0;
            for (int64_t e = c->h_ptr[r]; e < c->h_ptr[r + 1]; e++) prod *= tanh(bits[e] / 2.0);
            products[r] = prod;
        }
        for (uint32_t b = 0; b < n; b++)
            for (int64_t k = c->t_ptr[b]; k < c->t_ptr[b + 1]; k++) {
                uint32_t r = c->t_idx[k];
                double inner = products[r] / tanh(bits[c->t_edge[k]] / 2.0);
                double v = 2.0 * rust_atanh(inner);
                if (getbit(syn, r)) v *= -1.0;
                checks[k] = v;
            }
        /* update_bits :168-180 */
        for (uint32_t b = 0; b < n; b++) {
            double s = 0.0;
            for (int64_t k = c->t_ptr[b]; k < c->t_ptr[b + 1]; k++) s += checks[k];
            sums[b] = s;
        }
        for (uint32_t r = 0; r < m; r++)
            for (int64_t e = c->h_ptr[r]; e < c->h_ptr[r + 1]; e++) {
                uint32_t b = c->h_idx[e];
                bits[e] = sums[b] - checks[c->h_csc[e]] + prior;
            }
    }
    memcpy(corr, dec, nwords(n) * 8);
    if (iters) *iters = it;
    if (post) memcpy(post, lik, sizeof(double) * n);
    if (chk) for (int64_t e = 0; e < E; e++) chk[e] = checks[c->h_csc[e]];
    free(bits); free(checks); free(products); free(sums); free(lik); free(dec);
}

/* LinearDecoder::decode (belief_propagation.rs:15-19): message ^ correction_for(H message) */
void orc_bp_decode_words(const orc_code *c, double p, uint32_t max_iter, const uint64_t *msg,
                         uint64_t *out, uint32_t *iters)
{
    size_t wn = nwords(c->n), wm = nwords(c->m);
    uint64_t *s = (uint64_t *)malloc((wm ? wm : 1) * 8);
    orc_syndrome_words(c, msg, s);
    orc_bp_correction_for(c, p, max_iter, s, out, iters, NULL, NULL);
    for (size_t w = 0; w < wn; w++) out[w] ^= msg[w];
    free(s);
}

/* orc_bp_correction_for over a batch of syndromes, shots spread over OpenMP threads (the wide GPU parity tests):
 * syn [nshots][wm], corr [nshots][wn], iters [nshots], post [nshots][n] or NULL, chk [nshots][E] or NULL. */
void orc_bp_correction_for_batch(const orc_code *c, double p, uint32_t max_iter, uint64_t nshots, const uint64_t *syn,
                                 uint64_t *corr, uint32_t *iters, double *post, double *chk, int nthreads)
{
    size_t wn = nwords(c->n) ? nwords(c->n) : 1, wm = nwords(c->m) ? nwords(c->m) : 1;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < (int64_t)nshots; i++)
        orc_bp_correction_for(c, p, max_iter, syn + (size_t)i * wm, corr + (size_t)i * wn, iters + i,
                              post ? post + (size_t)i * c->n : NULL, chk ? chk + (size_t)i * (size_t)c->E : NULL);
}

/* ------------------------------------------------------------------------ */
/* Noise.  Distribution per binary_symmetric_channel.rs:35-43 and            */
/* depolarizing.rs:34-48; the stream is this project's specification         */
/* (docs: DESIGN.md "Noise specification"), identical on CPU and GPU.        */
/* ------------------------------------------------------------------------ */

/* Philox4x32-10 (Salmon et al., SC'11), counter (c0..c3), key (k0,k1). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
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

/* Gap table: Q[j] = min(floor(q^j * 2^32), 2^32-1), q^j by repeated double
 * multiplication, j = 1..64.  P(gap >= j) = Q[j] / 2^32. */
void orc_gap_table(double p, uint32_t Q[65])
{
    double q = 1.0 - p, qj = 1.0;
    Q[0] = 0xFFFFFFFFu;
    for (int j = 1; j <= 64; j++) {
        qj = qj * q;
        double t = floor(qj * 4294967296.0);
        Q[j] = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    }
}

typedef struct {
    uint32_t ctr[4], key[2], buf[4];
    int have;
} draw_stream;

static void stream_init(draw_stream *s, uint64_t seed, uint64_t shot, uint32_t word)
{
    s->key[0] = (uint32_t)seed; s->key[1] = (uint32_t)(seed >> 32);
    s->ctr[0] = (uint32_t)shot; s->ctr[1] = (uint32_t)(shot >> 32);
    s->ctr[2] = word; s->ctr[3] = 0;
    s->have = 0;
}

static uint32_t stream_draw(draw_stream *s)
{
    if (!s->have) {
        orc_philox4x32_10(s->ctr, s->key, s->buf);
        s->ctr[3]++;
        s->have = 4;
    }
    uint32_t u = s->buf[4 - s->have];
    s->have--;
    return u;
}

static uint32_t gap_of(const uint32_t Q[65], uint32_t u)
{
    uint32_t g = 0;
    for (int j = 1; j <= 64; j++) { if (u < Q[j]) g++; else break; }
    return g;
}

/* One 64-bit word of a BSC(p) sample: each bit i.i.d. Bernoulli(p)
 * (binary_symmetric_channel.rs:35-43), drawn as geometric gaps. */
static uint64_t bsc_word(const uint32_t Q[65], double p, uint64_t seed, uint64_t shot, uint32_t w,
                         uint32_t nbits)
{
    if (p == 0.0) return 0;
    draw_stream s; stream_init(&s, seed, shot, w);
    uint64_t word = 0; uint32_t pos = 0;
    for (;;) {
        pos += gap_of(Q, stream_draw(&s));
        if (pos >= nbits) break;
        word |= (uint64_t)1 << pos;
        pos++;
        if (pos >= nbits) break;
    }
    return word;
}

void orc_sample_bsc_words(uint64_t seed, uint64_t shot, double p, uint32_t n, uint64_t *words)
{
    uint32_t Q[65]; orc_gap_table(p, Q);
    size_t W = nwords(n);
    for (size_t w = 0; w < W; w++) {
        uint32_t nb = (uint32_t)((size_t)n - w * 64 < 64 ? (size_t)n - w * 64 : 64);
        words[w] = bsc_word(Q, p, seed, shot, (uint32_t)w, nb);
    }
}

/* Depolarizing noise (depolarizing.rs:34-48): each qubit is hit with
 * probability p, a hit picks X, Y or Z uniformly.  x-part = X or Y positions,
 * z-part = Z or Y positions (css.rs:92-99). */
static void depol_word(const uint32_t Q[65], double p, uint64_t seed, uint64_t shot, uint32_t w,
                       uint32_t nbits, uint64_t *xw, uint64_t *zw)
{
    *xw = 0; *zw = 0;
    if (p == 0.0) return;
    draw_stream s; stream_init(&s, seed, shot, w);
    uint32_t pos = 0;
    for (;;) {
        pos += gap_of(Q, stream_draw(&s));
        if (pos >= nbits) break;
        uint32_t pauli = (uint32_t)(((uint64_t)stream_draw(&s) * 3u) >> 32); /* 0=X 1=Y 2=Z */
        if (pauli <= 1) *xw |= (uint64_t)1 << pos;
        if (pauli >= 1) *zw |= (uint64_t)1 << pos;
        pos++;
        if (pos >= nbits) break;
    }
}

void orc_sample_depolarizing_words(uint64_t seed, uint64_t shot, double p, uint32_t n,
                                   uint64_t *xw, uint64_t *zw)
{
    uint32_t Q[65]; orc_gap_table(p, Q);
    size_t W = nwords(n);
    for (size_t w = 0; w < W; w++) {
        uint32_t nb = (uint32_t)((size_t)n - w * 64 < 64 ? (size_t)n - w * 64 : 64);
        depol_word(Q, p, seed, shot, (uint32_t)w, nb, &xw[w], &zw[w]);
    }
}

/* ------------------------------------------------------------------------ */
/* CSS glue.                                                                 */
/* ------------------------------------------------------------------------ */

static int any_nonzero(const uint64_t *w, size_t n) { for (size_t i = 0; i < n; i++) if (w[i]) return 1; return 0; }

/* CssCode::syndrome_of (css/mod.rs:200-205): s.x = Hx * e_z, s.z = Hz * e_x */
void orc_css_syndrome_words(const orc_code *hx, const orc_code *hz, const uint64_t *ex, const uint64_t *ez,
                            uint64_t *sx, uint64_t *sz)
{
    orc_syndrome_words(hx, ez, sx);
    orc_syndrome_words(hz, ex, sz);
}

/* !CssCode::has_stabilizer(residual) (css/mod.rs:225-252), residual given by
 * its x and z parts.  has_logical: both syndromes trivial; then every logical
 * generator must commute: X-type generators (rows of lx) with the z part,
 * Z-type generators (rows of lz) with the x part.
 * Returns bit0: x part fails (Hz rx != 0 or Lz rx != 0), bit1: z part fails. */
int orc_css_residual_fails(const orc_code *hx, const orc_code *hz, const orc_code *lx, const orc_code *lz,
                           const uint64_t *rx, const uint64_t *rz)
{
    size_t big = nwords(hx->m);
    if (nwords(hz->m) > big) big = nwords(hz->m);
    if (nwords(lx->m) > big) big = nwords(lx->m);
    if (nwords(lz->m) > big) big = nwords(lz->m);
    uint64_t *t = (uint64_t *)malloc((big ? big : 1) * 8);
    int fx = 0, fz = 0;
    orc_syndrome_words(hz, rx, t); fx |= any_nonzero(t, nwords(hz->m));
    orc_syndrome_words(lz, rx, t); fx |= any_nonzero(t, nwords(lz->m));
    orc_syndrome_words(hx, rz, t); fz |= any_nonzero(t, nwords(hx->m));
    orc_syndrome_words(lx, rz, t); fz |= any_nonzero(t, nwords(lx->m));
    free(t);
    return fx | (fz << 1);
}

/* ------------------------------------------------------------------------ */
/* Monte-Carlo drivers (the loop of examples/css_decoding.rs:14-21 and the   */
/* classical shot of SURVEY.md 3.4).  counters: see include/ldpc_hip.h.      */
/* ------------------------------------------------------------------------ */
enum { CNT_SHOTS = 0, CNT_FAIL = 1, CNT_DETECTED = 2, CNT_UNDETECTED = 3, CNT_WORK = 4,
       CNT_CAPPED = 5, CNT_XFAIL = 6, CNT_ZFAIL = 7, CNT_N = 8 };

static void add_counters(uint64_t *dst, const uint64_t *src)
{
    for (int i = 0; i < CNT_N; i++) {
#ifdef _OPENMP
#pragma omp atomic
#endif
        dst[i] += src[i];
    }
}

/* flip decoder, BSC noise, all-zero codeword.  reference_style != 0 uses the
 * decoder restated as written (rescans), else the incremental one. */
void orc_mc_flip(const orc_code *c, double p, uint64_t seed, uint64_t first, uint64_t nshots,
                 int reference_style, int nthreads, uint64_t *counters)
{
    memset(counters, 0, sizeof(uint64_t) * CNT_N);
    size_t wn = nwords(c->n), wm = nwords(c->m);
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
    {
        uint64_t loc[CNT_N] = {0};
        uint64_t *e = (uint64_t *)malloc((wn ? wn : 1) * 8), *o = (uint64_t *)malloc((wn ? wn : 1) * 8);
        uint64_t *s = (uint64_t *)malloc((wm ? wm : 1) * 8);
        uint32_t *pi = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)c->n + 1));
        uint32_t *po = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)c->n + 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int64_t i = 0; i < (int64_t)nshots; i++) {
            uint64_t shot = first + (uint64_t)i, flips = 0;
            orc_sample_bsc_words(seed, shot, p, c->n, e);
            if (reference_style) {
                size_t np = orc_words_to_pos(e, c->n, pi);
                int64_t no = orc_flip_decode_pos(c, c->n, pi, np, po, &flips);
                orc_pos_to_words(po, (size_t)no, c->n, o);
            } else {
                orc_flip_decode_words(c, e, o, NULL, &flips);
            }
            orc_syndrome_words(c, o, s);
            int fail = any_nonzero(o, wn), det = any_nonzero(s, wm);
            loc[CNT_SHOTS]++; loc[CNT_FAIL] += (uint64_t)fail; loc[CNT_DETECTED] += (uint64_t)det;
            loc[CNT_UNDETECTED] += (uint64_t)(fail && !det); loc[CNT_WORK] += flips;
        }
        add_counters(counters, loc);
        free(e); free(o); free(s); free(pi); free(po);
    }
}

/* BP decoder, BSC noise, all-zero codeword: residual = e ^ correction_for(H e). */
void orc_mc_bp(const orc_code *c, double p, uint32_t max_iter, uint64_t seed, uint64_t first,
               uint64_t nshots, int nthreads, uint64_t *counters)
{
    memset(counters, 0, sizeof(uint64_t) * CNT_N);
    size_t wn = nwords(c->n), wm = nwords(c->m);
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
    {
        uint64_t loc[CNT_N] = {0};
        uint64_t *e = (uint64_t *)malloc((wn ? wn : 1) * 8), *o = (uint64_t *)malloc((wn ? wn : 1) * 8);
        uint64_t *s = (uint64_t *)malloc((wm ? wm : 1) * 8);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int64_t i = 0; i < (int64_t)nshots; i++) {
            uint64_t shot = first + (uint64_t)i; uint32_t it = 0;
            orc_sample_bsc_words(seed, shot, p, c->n, e);
            orc_syndrome_words(c, e, s);
            orc_bp_correction_for(c, p, max_iter, s, o, &it, NULL, NULL);
            for (size_t w = 0; w < wn; w++) o[w] ^= e[w];
            orc_syndrome_words(c, o, s);
            int fail = any_nonzero(o, wn), det = any_nonzero(s, wm);
            loc[CNT_SHOTS]++; loc[CNT_FAIL] += (uint64_t)fail; loc[CNT_DETECTED] += (uint64_t)det;
            loc[CNT_UNDETECTED] += (uint64_t)(fail && !det); loc[CNT_WORK] += it;
            loc[CNT_CAPPED] += (uint64_t)det;
        }
        add_counters(counters, loc);
        free(e); free(o); free(s);
    }
}

/* examples/css_decoding.rs:14-21 with CssDecoder<BpDecoder> (decoders/css.rs:10-15):
 * correction.z = BP_Hx(s.x), correction.x = BP_Hz(s.z); both decoders use the
 * channel probability as prior (css_decoding.rs:42-43). */
void orc_mc_css_bp(const orc_code *hx, const orc_code *hz, const orc_code *lx, const orc_code *lz,
                   double p, uint32_t max_iter, uint64_t seed, uint64_t first, uint64_t nshots,
                   int nthreads, uint64_t *counters)
{
    memset(counters, 0, sizeof(uint64_t) * CNT_N);
    size_t wn = nwords(hx->n), wmx = nwords(hx->m), wmz = nwords(hz->m);
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
    {
        uint64_t loc[CNT_N] = {0};
        uint64_t *ex = (uint64_t *)malloc((wn ? wn : 1) * 8), *ez = (uint64_t *)malloc((wn ? wn : 1) * 8);
        uint64_t *cx = (uint64_t *)malloc((wn ? wn : 1) * 8), *cz = (uint64_t *)malloc((wn ? wn : 1) * 8);
        uint64_t *sx = (uint64_t *)malloc((wmx ? wmx : 1) * 8), *sz = (uint64_t *)malloc((wmz ? wmz : 1) * 8);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int64_t i = 0; i < (int64_t)nshots; i++) {
            uint64_t shot = first + (uint64_t)i; uint32_t itx = 0, itz = 0;
            orc_sample_depolarizing_words(seed, shot, p, hx->n, ex, ez);
            orc_css_syndrome_words(hx, hz, ex, ez, sx, sz);
            orc_bp_correction_for(hx, p, max_iter, sx, cz, &itx, NULL, NULL);
            orc_bp_correction_for(hz, p, max_iter, sz, cx, &itz, NULL, NULL);
            for (size_t w = 0; w < wn; w++) { cx[w] ^= ex[w]; cz[w] ^= ez[w]; }
            int f = orc_css_residual_fails(hx, hz, lx, lz, cx, cz);
            loc[CNT_SHOTS]++; loc[CNT_FAIL] += (uint64_t)(f != 0);
            loc[CNT_WORK] += (uint64_t)itx + itz;
            loc[CNT_XFAIL] += (uint64_t)(f & 1); loc[CNT_ZFAIL] += (uint64_t)((f >> 1) & 1);
        }
        add_counters(counters, loc);
        free(ex); free(ez); free(cx); free(cz); free(sx); free(sz);
    }
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
