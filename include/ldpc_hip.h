/*
 * ldpc_hip.h -- C ABI of the MI355X (gfx950) Monte-Carlo decoding path of the
 * `ldpc` crate (maxtremblay/ldpc 0.7.0).
 *
 * This is the drop-in boundary: exactly the entry points a Rust FFI shim
 * (`extern "C"` block + cc/bindgen, see INTEGRATION.md and bindings/rust/)
 * binds to implement the crate's own traits and types on the GPU.  Every
 * declaration cites the reference interface it replaces (paths relative to the
 * reference repository).
 *
 * Conventions
 *  - Every function returns an ldpc_status (0 = ok).  ldpc_last_error() gives
 *    the message of the last failure on the calling thread.  Nothing unwinds
 *    across the ABI.  Where the reference panics (length mismatch, probability
 *    outside [0,1], unsorted rows) the call returns LDPC_ERR_BAD_ARG or
 *    LDPC_ERR_LENGTH_MISMATCH and the shim turns that into the same panic.
 *  - Sparse binary vectors cross the boundary as SparseBinVec does: a length
 *    and a strictly ascending `size_t` position list.  The caller owns every
 *    buffer; output position buffers need room for `len` entries.
 *  - Batched entry points take DEVICE pointers to bit-packed vectors: shot-major
 *    u64 words, bit i of a vector in word i/64 at bit i%64, ceil(len/64) words
 *    per shot.
 *  - Noise is counter based (Philox4x32-10, key = seed, counter = (shot, word,
 *    draw block)); a shot is identified by its 64-bit global index, so results
 *    do not depend on batch size, launch geometry or the number of GPUs.
 *  - All compute (sampling, syndromes, decoding, failure tests) runs in HIP
 *    kernels on the current device; there is no CPU fallback.  Handles are
 *    bound to the device that was current when they were created.  Calls on one
 *    handle are serialized internally; different handles may be used from
 *    different threads.
 */
#ifndef LDPC_HIP_H
#define LDPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ldpc_status {
    LDPC_OK = 0,
    LDPC_ERR_BAD_ARG = 1,          /* reference: panic on malformed input            */
    LDPC_ERR_LENGTH_MISMATCH = 2,  /* reference: panic, linear_code/mod.rs:397-403   */
    LDPC_ERR_SAMPLING = 3,         /* reference: SamplingError, random.rs:78-95      */
    LDPC_ERR_CSS = 4,              /* reference: CssError, css/mod.rs:366-381        */
    LDPC_ERR_HIP = 5,              /* HIP runtime failure / no GPU                   */
    LDPC_ERR_CAPACITY = 6          /* problem exceeds what the kernels were sized for */
} ldpc_status;

/* Monte-Carlo counters (u64[LDPC_COUNTERS]); sums over shots, safe to add
 * across GPUs (bench.py all-reduces them over RCCL). */
enum {
    LDPC_CNT_SHOTS = 0,      /* shots processed                                          */
    LDPC_CNT_FAILURES = 1,   /* classical: residual != 0; CSS: !has_stabilizer(residual)  */
    LDPC_CNT_DETECTED = 2,   /* classical: H*residual != 0                                */
    LDPC_CNT_UNDETECTED = 3, /* classical: residual != 0 and H*residual == 0              */
    LDPC_CNT_WORK = 4,       /* flip decoder: flips; BP: iterations (CSS: x + z)          */
    LDPC_CNT_CAPPED = 5,     /* BP: stopped by the iteration cap without syndrome match   */
    LDPC_CNT_XFAIL = 6,      /* CSS: x part of the residual not a stabilizer              */
    LDPC_CNT_ZFAIL = 7,      /* CSS: z part of the residual not a stabilizer              */
    LDPC_COUNTERS = 8
};

typedef struct ldpc_code ldpc_code;                 /* LinearCode        codes/linear_code/mod.rs:52-56 */
typedef struct ldpc_css_code ldpc_css_code;         /* CssCode           codes/css/mod.rs:25-28         */
typedef struct ldpc_flip_decoder ldpc_flip_decoder; /* FlipDecoder<Code> decoders/flip.rs:7-9           */
typedef struct ldpc_bp_decoder ldpc_bp_decoder;     /* BpDecoder         decoders/belief_propagation.rs:8-12 */
typedef struct ldpc_css_bp_decoder ldpc_css_bp_decoder; /* CssDecoder<BpDecoder> decoders/css.rs:4 */

/* ---- runtime ---------------------------------------------------------- */
const char *ldpc_last_error(void);
int ldpc_device_count(int *count);
int ldpc_set_device(int device);
/* Work is enqueued on this hipStream_t (default: the null stream) and every
 * entry point that returns host data synchronizes it before returning. */
int ldpc_set_stream(void *hip_stream);
/* Duration of the decoding kernels of the most recent call on this thread,
 * measured with hipEvents on the launch stream (bench.py roofline). */
double ldpc_last_kernel_ms(void);
/* Name of the decoding kernel this thread launched last (the library picks among several forms of a decoder by
 * code and batch size; bench.py names the one it timed).  No counterpart in the reference. */
const char *ldpc_last_kernel_name(void);
/* BP iterations that the calling thread's last Monte-Carlo call (ldpc_bp_mc_run, ldpc_css_bp_mc_run: both halves)
 * actually ran.  The counters report what belief_propagation.rs:23-30 would have run (LDPC_CNT_WORK); the LDS form of
 * the decoder runs fewer when a run that never converges is found to repeat itself with period two, which decides its
 * result at max_iterations exactly.  bench.py prices its roofline on this number.  No counterpart in the reference. */
uint64_t ldpc_last_bp_executed_iterations(void);
/* The LDPC_HIP_* environment variables (A/B, test and profiling switches; none is needed in production) are read
 * once, when the library first needs one.  A test that changes one afterwards calls this to have them read again.
 * No counterpart in the reference (it has no configuration).  Not to be called while other
 * threads are inside decoding calls. */
void ldpc_debug_reload_switches(void);

/* ---- LinearCode (codes/linear_code/mod.rs) ----------------------------- */
/* SparseBinMat::new(n_bits, rows) + LinearCode::from_parity_check_matrix
 * (mod.rs:90-98).  Rows are CSR: row_ptr[n_checks+1], col_idx[row_ptr[n_checks]],
 * strictly ascending inside a row.  The generator matrix is NOT computed (its
 * nullspace is host algebra off the decoding path and infeasible at n = 1e6). */
int ldpc_code_create(size_t n_bits, size_t n_checks, const size_t *row_ptr, const size_t *col_idx,
                     ldpc_code **out);
/* LinearCode::from_generator_matrix (mod.rs:116-124): the parity checks are a basis of the nullspace of the
 * generators (host elimination, small codes; the basis is this library's, see ldpc_code_generator_matrix).
 * LinearCode::from_both_matrices (mod.rs:59-72): LDPC_ERR_BAD_ARG "matrices are non orthogonal" for the panic of
 * :64-66; the generators are checked and not kept. */
int ldpc_code_from_generator_matrix(size_t n_bits, size_t n_generators, const size_t *row_ptr, const size_t *col_idx,
                                    ldpc_code **out);
int ldpc_code_from_both_matrices(size_t n_bits, size_t n_generators, const size_t *g_row_ptr, const size_t *g_col_idx,
                                 size_t n_checks, const size_t *h_row_ptr, const size_t *h_col_idx, ldpc_code **out);
int ldpc_code_hamming(ldpc_code **out);                     /* mod.rs:161-167 */
int ldpc_code_repetition(size_t length, ldpc_code **out);   /* mod.rs:139-143 */
/* LinearCode::random_regular_code()...sample_with (mod.rs:202-204, random.rs:56-76).
 * LDPC_ERR_SAMPLING iff n_bits*bit_degree != n_checks*check_degree (or no simple
 * graph exists).  The graph is drawn by this library's own seeded sampler: the
 * reference delegates to the un-vendored `bigs` crate, so graph identity is
 * unpinned; only the contract (simple, regular, rows ascending) is kept. */
int ldpc_code_random_regular(size_t n_bits, size_t n_checks, size_t bit_degree, size_t check_degree,
                             uint64_t seed, ldpc_code **out);
size_t ldpc_code_len(const ldpc_code *code);                /* mod.rs:278-280 */
size_t ldpc_code_num_checks(const ldpc_code *code);         /* mod.rs:284-286 */
size_t ldpc_code_num_edges(const ldpc_code *code);          /* number_of_ones */
/* parity_check_matrix() / bit_adjacencies() (mod.rs:207-209, 236-238) as CSR */
int ldpc_code_parity_check_matrix(const ldpc_code *code, size_t *row_ptr, size_t *col_idx);
int ldpc_code_bit_adjacencies(const ldpc_code *code, size_t *row_ptr, size_t *col_idx);
/* edges() (mod.rs:366-368, edges.rs:23-29): row-major (check, bit) pairs */
int ldpc_code_edges(const ldpc_code *code, size_t *bits, size_t *checks);
/* syndrome_of / has_codeword (mod.rs:393-405, 431-436) */
int ldpc_code_syndrome_of(const ldpc_code *code, size_t len, const size_t *positions, size_t n_positions,
                          size_t *out_positions, size_t *out_n);
int ldpc_code_has_codeword(const ldpc_code *code, size_t len, const size_t *positions, size_t n_positions,
                           int *out);
/* batched H*e on device words: d_words [n_shots][ceil(n/64)] -> d_syndromes [n_shots][ceil(m/64)] */
int ldpc_code_syndrome_batch(const ldpc_code *code, size_t n_shots, const uint64_t *d_words,
                             uint64_t *d_syndromes);
/* Host algebra of LinearCode (small codes: dense GF(2) elimination on the host, not part of the decoding path).
 * generator_matrix() (mod.rs:225-227) is a basis of the nullspace of H computed on demand; the reference's
 * basis comes from the un-vendored sparse-bin-mat crate, so only its row space is comparable.
 * num_generators (mod.rs:290-292), dimension (mod.rs:308-310), minimal_distance (mod.rs:318-335; *found = 0
 * when the code has no non-trivial codeword; LDPC_ERR_CAPACITY above 30 generators), has_same_codespace
 * (mod.rs:272-275). */
int ldpc_code_num_generators(const ldpc_code *code, size_t *out);
int ldpc_code_generator_matrix(const ldpc_code *code, size_t *row_ptr, size_t *col_idx, size_t col_idx_capacity);
int ldpc_code_dimension(const ldpc_code *code, size_t *out);
int ldpc_code_minimal_distance(const ldpc_code *code, size_t *distance, int *found);
int ldpc_code_has_same_codespace(const ldpc_code *code, const ldpc_code *other, int *out);
/* LinearCode::as_json (mod.rs:468-471) and its inverse, serde_json::from_str::<LinearCode> (derive mod.rs:51):
 * {"parity_check_matrix":M,"generator_matrix":M,"bit_adjacencies":M}.  PARITY UNPINNED: M is the serde form of the
 * un-vendored sparse-bin-mat crate's SparseBinMat, assumed {"row_ranges":[..],"column_indices":[..],
 * "number_of_columns":n}; the reference holds no sample of it.  as_json writes a NUL-terminated string of
 * *needed bytes (NUL included) when capacity allows, else returns LDPC_ERR_CAPACITY with *needed set (call with
 * capacity 0 to size the buffer).  from_json needs parity_check_matrix; the other members, when present, must agree
 * with it (transpose; orthogonal, as from_both_matrices mod.rs:59-72 demands) or the call returns LDPC_ERR_BAD_ARG. */
int ldpc_code_as_json(const ldpc_code *code, char *buf, size_t capacity, size_t *needed);
int ldpc_code_from_json(const char *text, size_t len, ldpc_code **out);
void ldpc_code_destroy(ldpc_code *code);

/* ---- noise (src/noise) ------------------------------------------------- */
/* BinarySymmetricChannel::with_probability(p).sample_error_of_length(len, rng)
 * (binary_symmetric_channel.rs:22-43); (seed, shot) stand for the rng. */
int ldpc_bsc_sample_error_of_length(double probability, size_t len, uint64_t seed, uint64_t shot,
                                    size_t *out_positions, size_t *out_n);
/* DepolarizingNoise (depolarizing.rs:20-48): out_paulis[i] in {1=X, 2=Y, 3=Z}. */
int ldpc_depolarizing_sample_error_of_length(double probability, size_t len, uint64_t seed, uint64_t shot,
                                             size_t *out_positions, uint8_t *out_paulis, size_t *out_n);
/* LinearCode::random_error (mod.rs:460-466) / CssCode::random_error
 * (css/mod.rs:357-363), batched on device: shots first_shot .. first_shot+n_shots-1.
 * The depolarizing form writes the x part (X or Y) and z part (Z or Y) of each
 * operator (css.rs:92-99). */
int ldpc_bsc_sample_batch(double probability, size_t len, uint64_t seed, uint64_t first_shot, size_t n_shots,
                          uint64_t *d_words);
/* ErasureChannel::with_probability(p).sample_error_of_length(len, rng) (noise/erasure.rs:19-43): the
 * positions of the erased bits / qubits; the same i.i.d. Bernoulli(p) stream as the binary symmetric channel. */
int ldpc_erasure_sample_error_of_length(double probability, size_t len, uint64_t seed, uint64_t shot,
                                        size_t *out_positions, size_t *out_n);
int ldpc_depolarizing_sample_batch(double probability, size_t len, uint64_t seed, uint64_t first_shot,
                                   size_t n_shots, uint64_t *d_x_words, uint64_t *d_z_words);

/* ---- FlipDecoder (decoders/flip.rs) ------------------------------------ */
int ldpc_flip_create(const ldpc_code *code, ldpc_flip_decoder **out);              /* flip.rs:12-14 */
/* FlipDecoder::decode (flip.rs:21-33): returns the corrected WORD */
int ldpc_flip_decode(ldpc_flip_decoder *dec, size_t len, const size_t *positions, size_t n_positions,
                     size_t *out_positions, size_t *out_n);
/* batched decode; outputs may be NULL: d_syndromes_out = H*input (initial
 * syndrome), d_flips_out = flips per shot */
int ldpc_flip_decode_batch(ldpc_flip_decoder *dec, size_t n_shots, const uint64_t *d_words_in,
                           uint64_t *d_words_out, uint64_t *d_syndromes_out, uint32_t *d_flips_out);
/* Monte-Carlo: sample BSC(p) error -> syndrome -> decode -> failure test, all on
 * device.  Optional record buffers (device, may be NULL) receive
 * the sampled errors, their syndromes and the decoded words. */
int ldpc_flip_mc_run(ldpc_flip_decoder *dec, double probability, uint64_t seed, uint64_t first_shot,
                     size_t n_shots, uint64_t counters[LDPC_COUNTERS], uint64_t *d_errors_out,
                     uint64_t *d_syndromes_out, uint64_t *d_words_out);
/* ldpc_flip_mc_run for callers that cannot allocate device memory: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference (which has no device).  Same results as
 * ldpc_flip_mc_run, bit for bit (the one-wave kernels serve it whatever the size of the run, and the diagnostic
 * flip limit is ignored), but every piece of device scratch (counters, parked input words, flip logs) comes
 * from a static device array inside the library (64 MiB per GPU, held until the process ends): no hipMalloc, no
 * hipFree, no event, no synchronisation, and only stream operations (one memset, one launch, one copy; a refused call
 * queues nothing).  The eight counters are copied to d_counters_out (DEVICE memory, [LDPC_COUNTERS] u64) on the
 * library's stream; they and the optional outputs are valid once that stream has been synchronised.  The handle must
 * have served one ordinary call before (its graph and tables are uploaded then, and the array's address on that GPU is
 * looked up then).
 * LDPC_ERR_CAPACITY when a code is too large for the static array (a run is otherwise only slowed: fewer blocks).
 * Calls on one GPU share the array: two that overlap in time (two streams, two threads) race. */
int ldpc_flip_mc_run_noalloc(ldpc_flip_decoder *dec, double probability, uint64_t seed, uint64_t first_shot,
                             size_t n_shots, uint64_t *d_counters_out, uint64_t *d_errors_out,
                             uint64_t *d_syndromes_out, uint64_t *d_words_out);
void ldpc_flip_destroy(ldpc_flip_decoder *dec);

/* ---- BpDecoder (decoders/belief_propagation.rs) ------------------------ */
/* BpDecoder::new(&H, Probability, num_iterations) (belief_propagation.rs:36-46) */
int ldpc_bp_create(const ldpc_code *code, double probability, size_t max_iterations, ldpc_bp_decoder **out);
/* SyndromeDecoder::correction_for (belief_propagation.rs:23-30) */
int ldpc_bp_correction_for(ldpc_bp_decoder *dec, size_t syndrome_len, const size_t *syndrome_positions,
                           size_t n_positions, size_t *out_positions, size_t *out_n);
/* LinearDecoder::decode (belief_propagation.rs:15-19) */
int ldpc_bp_decode(ldpc_bp_decoder *dec, size_t len, const size_t *positions, size_t n_positions,
                   size_t *out_positions, size_t *out_n);
/* batched forms; d_iterations_out [n_shots] and d_posteriors_out [n_shots][n]
 * (posterior LLRs at stop, f64) may be NULL */
int ldpc_bp_correction_for_batch(ldpc_bp_decoder *dec, size_t n_shots, const uint64_t *d_syndromes,
                                 uint64_t *d_corrections_out, uint32_t *d_iterations_out,
                                 double *d_posteriors_out);
/* The same with the check -> bit messages at stop as a further output: d_check_messages_out [n_shots][E] f64 (may be
 * NULL), edge order = Edges (edges.rs:23-29); the values `Messages::update_checks` last wrote
 * (belief_propagation.rs:156-163), 0.0 for a shot that stopped before its first update (:68-76).  A parity / debug
 * output: with max_iterations = 1, 2, ... it shows the messages after each iteration. */
int ldpc_bp_correction_for_batch_messages(ldpc_bp_decoder *dec, size_t n_shots, const uint64_t *d_syndromes,
                                          uint64_t *d_corrections_out, uint32_t *d_iterations_out,
                                          double *d_posteriors_out, double *d_check_messages_out);
int ldpc_bp_decode_batch(ldpc_bp_decoder *dec, size_t n_shots, const uint64_t *d_words_in,
                         uint64_t *d_words_out, uint32_t *d_iterations_out);
/* Monte-Carlo with BSC(noise_probability) errors; the decoder keeps the prior it
 * was created with.  d_failed_out (device, u8 per shot, may be NULL) is set to 1
 * for failed shots. */
int ldpc_bp_mc_run(ldpc_bp_decoder *dec, double noise_probability, uint64_t seed, uint64_t first_shot,
                   size_t n_shots, uint64_t counters[LDPC_COUNTERS], uint8_t *d_failed_out);
void ldpc_bp_destroy(ldpc_bp_decoder *dec);

/* ---- CssCode (codes/css/mod.rs, css.rs) -------------------------------- */
/* CssCode::try_new (css/mod.rs:35-50): LDPC_ERR_CSS on different lengths or
 * Hx*Hz^T != 0.  Logical generators by the reference's procedure
 * (css/logicals.rs:39-85) on a nullspace basis computed here; small codes only. */
int ldpc_css_create(const ldpc_code *x_code, const ldpc_code *z_code, ldpc_css_code **out);
/* explicit stabilizers and logicals, as CssCode::shor_code builds them (css/mod.rs:60-81) */
int ldpc_css_create_with_logicals(const ldpc_code *x_stabs, const ldpc_code *z_stabs,
                                  const ldpc_code *x_logicals, const ldpc_code *z_logicals,
                                  ldpc_css_code **out);
int ldpc_css_shor(ldpc_css_code **out);                                     /* css/mod.rs:60-81  */
int ldpc_css_steane(ldpc_css_code **out);                                   /* css/mod.rs:54-57  */
int ldpc_css_toric(size_t distance, ldpc_css_code **out);                   /* css/mod.rs:84-92  */
int ldpc_css_hypergraph_product(const ldpc_code *first, const ldpc_code *second,
                                ldpc_css_code **out);                       /* css/mod.rs:112-146 */
size_t ldpc_css_len(const ldpc_css_code *css);                              /* css/mod.rs:149-151 */
/* borrowed views, owned by the css handle: which = 0 x stabilizers, 1 z stabilizers,
 * 2 x logicals, 3 z logicals (css/mod.rs:256-276) */
const ldpc_code *ldpc_css_matrix(const ldpc_css_code *css, int which);
/* CssCode::syndrome_of (css/mod.rs:200-205): operator given by its x part (X or
 * Y positions) and z part (Z or Y positions); syndrome.x = Hx*z_part, syndrome.z = Hz*x_part */
int ldpc_css_syndrome_of(const ldpc_css_code *css, size_t len, const size_t *x_part, size_t n_x,
                         const size_t *z_part, size_t n_z, size_t *out_sx, size_t *out_n_sx,
                         size_t *out_sz, size_t *out_n_sz);
int ldpc_css_has_logical(const ldpc_css_code *css, size_t len, const size_t *x_part, size_t n_x,
                         const size_t *z_part, size_t n_z, int *out);       /* css/mod.rs:225-227 */
int ldpc_css_has_stabilizer(const ldpc_css_code *css, size_t len, const size_t *x_part, size_t n_x,
                            const size_t *z_part, size_t n_z, int *out);    /* css/mod.rs:247-252 */
/* CssErasureDecoder (decoders/css_erasure.rs:7-65; host rank algebra, small codes): the numbers of bad X and Z
 * errors supported on the erasure (:20-57) and ErasureDecoder::recovery_probability = 2^-(bad_x + bad_z) (:60-64) */
int ldpc_css_erasure_num_bad_errors(const ldpc_css_code *css, size_t len, const size_t *erasure, size_t n_erased,
                                    size_t *bad_x, size_t *bad_z);
int ldpc_css_erasure_recovery_probability(const ldpc_css_code *css, size_t len, const size_t *erasure,
                                          size_t n_erased, double *out);
/* serde form of CssCode (derive css/mod.rs:24-28; Css<T> css.rs:7-11): {"stabilizers":{"x":M,"z":M},
 * "logicals":{"x":M,"z":M}}, M as for ldpc_code_as_json (PARITY UNPINNED likewise).  from_json checks equal lengths
 * and Hx * Hz^T == 0 (css/mod.rs:36-42: LDPC_ERR_CSS) and keeps the logicals the text holds. */
int ldpc_css_as_json(const ldpc_css_code *css, char *buf, size_t capacity, size_t *needed);
int ldpc_css_from_json(const char *text, size_t len, ldpc_css_code **out);
void ldpc_css_destroy(ldpc_css_code *css);

/* ---- CssDecoder<BpDecoder> (decoders/css.rs, examples/css_decoding.rs) -- */
/* CssDecoder { x: BpDecoder::new(&Hx, p, it), z: BpDecoder::new(&Hz, p, it) }
 * (examples/css_decoding.rs:36-45) */
int ldpc_css_bp_create(const ldpc_css_code *css, double probability, size_t max_iterations,
                       ldpc_css_bp_decoder **out);
/* SyndromeDecoder<CssSyndromeView, CssOperator>::correction_for (decoders/css.rs:10-15):
 * correction.z = x_decoder(syndrome.x), correction.x = z_decoder(syndrome.z) */
int ldpc_css_bp_correction_for(ldpc_css_bp_decoder *dec, const size_t *sx, size_t n_sx, const size_t *sz,
                               size_t n_sz, size_t *out_cx, size_t *out_n_cx, size_t *out_cz,
                               size_t *out_n_cz);
/* The loop of examples/css_decoding.rs:14-21 on device: depolarizing(p) error ->
 * syndrome -> X and Z decodes -> !has_stabilizer(error * correction). */
int ldpc_css_bp_mc_run(ldpc_css_bp_decoder *dec, double noise_probability, uint64_t seed,
                       uint64_t first_shot, size_t n_shots, uint64_t counters[LDPC_COUNTERS]);
void ldpc_css_bp_destroy(ldpc_css_bp_decoder *dec);

/* ---- one process, several GPUs ------------------------------------------- */
/* The Monte-Carlo runs above, sharded over `n_devices` GPUs from ONE process (the route for a host that
 * is not launched one process per GPU, e.g. a Rust binary using the crate): one host thread per device,
 * each with its own decoder, decodes the global shot range [first_shot + n_shots*k/n, first_shot +
 * n_shots*(k+1)/n); the counters are summed on the host.  Shots are independent (examples/css_decoding.rs:14-21
 * carries only the failure count between iterations) and the noise is keyed by the global shot index, so the
 * result does not depend on n_devices.  devices = NULL means 0 .. n_devices-1; a device may be named twice. */
int ldpc_flip_mc_run_multi(const ldpc_code *code, int n_devices, const int *devices, double probability,
                           uint64_t seed, uint64_t first_shot, size_t n_shots, uint64_t counters[LDPC_COUNTERS]);
int ldpc_bp_mc_run_multi(const ldpc_code *code, int n_devices, const int *devices, double probability,
                         size_t max_iterations, uint64_t seed, uint64_t first_shot, size_t n_shots,
                         uint64_t counters[LDPC_COUNTERS]);
int ldpc_css_bp_mc_run_multi(const ldpc_css_code *css, int n_devices, const int *devices, double probability,
                             size_t max_iterations, uint64_t seed, uint64_t first_shot, size_t n_shots,
                             uint64_t counters[LDPC_COUNTERS]);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_HIP_H */
