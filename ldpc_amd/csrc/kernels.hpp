// Kernel argument blocks and host-side launchers (implemented in kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "device_common.hpp"

namespace ldpc {

// Diagnostic switches (the LDPC_HIP_* environment variables: A/B, test and profiling aids, never needed in
// production).  The environment is read ONCE, when the library first needs a switch; a test that changes a variable
// afterwards calls ldpc_debug_reload_switches() (not meant to run beside decoding calls of other threads).
struct Switches {
    int flip_kernel = 0;              // LDPC_HIP_FLIP_KERNEL: 0 default, 1 "csr" (generic kernel), 2 "v2" (flip2_kernel), 3 "v3" (flip3_kernel also for large runs)
    uint32_t flip_maxflips = 0xFFFFFFFFu;   // LDPC_HIP_FLIP_MAXFLIPS (flip2 profiling aid)
    int flip_lanes = 16;              // LDPC_HIP_FLIP_LANES (flip2: 16 or 8 lanes per shot)
    bool flip_prof = false;           // LDPC_HIP_FLIP_PROF: flip3 reports cycles per phase in spare counters
    int flip_bpc = 0;                 // LDPC_HIP_FLIP_BPC: flip3 blocks per CU (0: as many as LDS holds)
    int flip_nf = 0;                  // LDPC_HIP_FLIP_NF: flip4 flip waves of the 16 (0: default 8)
    int bp_waves = 0;                 // LDPC_HIP_BP_WAVES (0: by code size)
    int bp_oversub = 1;               // LDPC_HIP_BP_OVERSUB
    int bp_path = 0;                  // LDPC_HIP_BP_PATH: 0 by size, 1 "stream", 2 "persistent", 3 "lds" (where it serves), 4 "hbm" (never lds)
    long long bp_groups = 0;          // LDPC_HIP_BP_GROUPS (0: unset)
    bool bp_handoff_set = false;      // LDPC_HIP_BP_HANDOFF given
    uint32_t bp_handoff = 16;
    bool bp_profile = false;          // LDPC_HIP_BP_PROFILE
    uint32_t bp_hyb_kb = 0;           // LDPC_HIP_BP_HYB_KB (hybrid form: KB of LDS per workgroup; 0: 150)
    uint32_t bp_lds_threads = 0;      // LDPC_HIP_BP_LDS_THREADS (bp_lds_kernel's workgroup size; 0: chosen by bp_lds_shape)
    bool bp_stream_skew = false;      // LDPC_HIP_BP_STREAM_SKEW
    int bp_tail_burst = 0;            // LDPC_HIP_BP_TAIL_BURST (0: unset)
    bool bp_no_compact = false;       // LDPC_HIP_BP_NO_COMPACT
    int css_side_by_side = -1;        // LDPC_HIP_CSS_SIDE_BY_SIDE (-1: by size)
    bool debug = false;               // LDPC_HIP_DEBUG
};
const Switches &switches();
// name of the decoding kernel this thread launched last (bench.py's roofline object names it)
const char *last_launched_kernel();
void reload_switches();

enum NoiseSource : int {
    SRC_BSC = 0,          // sample BSC(p) on device
    SRC_DEPOL_X = 1,      // sample depolarizing(p), keep the x part (X or Y)
    SRC_DEPOL_Z = 2,      // sample depolarizing(p), keep the z part (Z or Y)
    SRC_WORDS = 3,        // read error / message words
    SRC_SYNDROMES = 4     // read syndromes (correction_for)
};

enum FailureTest : int {
    TEST_NONE = 0,
    TEST_RESIDUAL = 1,    // classical, all-zero codeword: residual != 0
    TEST_LOGICAL = 2      // CSS half: H*residual != 0 or L*residual != 0
};

struct FlipArgs {
    int source;                 // SRC_BSC or SRC_WORDS
    uint64_t seed, first_shot, n_shots;
    const uint64_t *words_in;   // [n_shots][Wn]       (SRC_WORDS)
    uint64_t *errors_out;       // [n_shots][Wn]  or null
    uint64_t *synd_out;         // [n_shots][Wm]  or null: H * input
    uint64_t *words_out;        // [n_shots][Wn]  or null: decoded word
    uint32_t *flips_out;        // [n_shots]      or null
    unsigned long long *counters;   // [LDPC_COUNTERS] device, accumulated
    uint32_t max_flips;         // diagnostics: stop a shot after this many flips (default: unlimited)
    uint64_t *o_scratch;        // fast kernel: [flip2_max_blocks][shots per wave][Wn] parked input words when errors_out is null
    uint32_t *flip_log;         // fast kernel: [flip2_max_blocks][shots per wave][flip_log_cap] flipped bits
    uint32_t flip_log_cap;      // >= m rounded up to 16, + 16
    uint32_t list_cap;          // fast kernels: entries of the LDS error list (flip2_list_cap / flip3_list_cap)
    double expected_weight;     // host only: mean weight of the channel's errors (sizes the list)
    GapTable gap;
};

// slot state of the streamed BP form (bp_stream_run), device memory: per group u64 masks over its 64 slots,
// per slot the shot index, iteration count and residual flag
struct BpsState {
    unsigned long long *alive;          // [2] per group set: active slots after the round's harvest (host loop condition)
    uint64_t *act, *fresh, *take, *mism; // [groups]
    uint64_t *myi;                      // [groups][64]
    uint32_t *iter, *flags;             // [groups][64]
    const uint32_t *gmap;               // when set: kernels work on the groups gmap[0..) instead of 0.. (tail)
    uint32_t *gmap_store;               // [groups] room for that list
    uint32_t *moves;                    // [groups][64] move table of the tail compaction (bps_compact)
};

struct BpArgs {
    int source;
    int test;
    int xor_input;              // decode(): output = input ^ correction
    uint32_t max_iter;
    uint32_t fail_bit;          // OR-ed into failed_out on failure
    uint32_t profile;           // diagnostics (LDPC_HIP_BP_PROFILE): block 0 prints its cycles per phase
    // Tail hand-off of the persistent kernel: once the shot queue is drained, a group whose active slots have all
    // run handoff_iter iterations or more stores its slot state in `handoff` and exits; bp_stream_run(resume)
    // finishes those shots with the whole chip.  handoff_iter = 0xFFFFFFFF: never.
    uint32_t handoff_iter;
    unsigned long long *handoff_flag;   // set to 1 by a group that hands off
    BpsState handoff;
    double prior;               // ln((1-p)/p), belief_propagation.rs:38
    uint64_t seed, first_shot, n_shots;
    unsigned long long *queue;  // device counter, zero at launch: shots handed out so far
    unsigned long long *executed;   // device counter or null: iterations actually run (bp_lds_kernel cuts periodic runs short)
    const uint64_t *words_in;   // [n_shots][Wn]
    const uint64_t *synd_in;    // [n_shots][Wm]
    uint64_t *corr_out;         // [n_shots][Wn] or null
    uint32_t *iters_out;        // [n_shots]     or null
    double *post_out;           // [n_shots][n]  or null
    double *chk_out;            // [n_shots][E]  or null: check -> bit messages at stop, Edges order (debug / parity)
    uint8_t *failed_out;        // [n_shots]     or null
    unsigned long long *counters;
    // logical test matrix (rows = logical generators), TEST_LOGICAL only
    uint32_t l_rows, l_nnz;
    const uint32_t *l_ptr, *l_idx;
    // workspace, one slice per shot group (64 shots)
    double *msg;                // [groups][2][E][64]
    uint64_t *ew;               // [groups][Wn][64] error words of the slot's shot
    uint64_t *dw;               // [groups][Wn][64] hard decisions, packed per slot
    uint64_t *dsl;              // [groups][n]      hard decisions, bit-sliced over slots
    uint64_t *ssl;              // [groups][m]      syndromes, bit-sliced over slots
    uint64_t *psl;              // [groups][m]      streamed form: syndrome ^ H * decision (the stop test's residual), bit-sliced
    uint64_t *esl;              // [groups][n]      error words of the slots being refilled, bit-sliced (set-up scratch)
    GapTable gap;
};

// 2-hop neighbourhood table of the flip decoder: for every bit b, K slots of
// (bit, its checks): slot 0 is b itself, the others are the bits that share a
// check with b.  One flip touches exactly these bits, so one contiguous
// K*(1+dvp)*4-byte read per flip replaces the H^T -> H -> H^T pointer chase.
struct NbTable {
    const uint32_t *rec;   // [n][K][1 + dvp] u32, unused entries 0xFFFFFFFF
    uint32_t kshift;       // K = 1 << kshift, K >= 16
    uint32_t dvp;          // checks per slot: 3 or 7
    uint32_t regular;      // every bit has exactly dvp checks (no sentinels among a valid slot's checks)
};

// Neighbourhood table of the third flip form (flip3_kernel): regular codes of bit degree 3, at most 16 slots per
// row, at most 32 windows of 512 mask bits.  Row b = 16 records {bitp, chk0, chk1, chk2}; bitp = (byte offset of the
// bit's 32-bit mask word) << 16 | bit in that word, chk as in NbTable.  Padding slots and the extra row n name the
// spare mask bit (last bit of the last window) and the spare check m, which is never set.
struct Nb3Table {
    const uint4 *rec;      // [n + 1][16]
    uint32_t sw;           // u64 words of a shot's syndrome in LDS: (m + 64) / 64
    uint32_t windows;      // (n + 512) / 512, <= 32
};

size_t bp_workspace_bytes_per_group(const DeviceGraph &g);
size_t flip3_lds_bytes(const Nb3Table &t, uint32_t list_cap);
uint32_t flip3_list_cap(double expected_weight);
uint32_t flip3_max_blocks();
uint32_t device_cu_count();
// block_cap (0: none): at most this many blocks (the scratch given in `a` is indexed by block)
hipError_t launch_flip3(const DeviceGraph &g, const Nb3Table &nb, const FlipArgs &a, bool prof, hipStream_t st, uint32_t block_cap = 0);
// the static device array behind ldpc_flip_mc_run_noalloc: its address on the current GPU and its size
hipError_t noalloc_scratch(unsigned char **ptr, size_t *bytes);
// fourth form (large runs): one 1024-thread workgroup per CU, flip waves + service waves
uint32_t flip4_slots(const Nb3Table &t, uint32_t list_cap, uint32_t nf);
uint32_t flip4_scratch_pieces();   // pieces (parked word / flip log) of HBM scratch a launch may index
hipError_t launch_flip4(const DeviceGraph &g, const Nb3Table &nb, const FlipArgs &a, hipStream_t st);
size_t flip2_lds_bytes(const DeviceGraph &g, uint32_t lanes_per_shot, uint32_t list_cap);
uint32_t flip2_list_cap(double expected_weight);
uint32_t flip2_lanes_per_shot(const DeviceGraph &g);
uint32_t flip2_max_blocks(const DeviceGraph &g);

hipError_t launch_sample_bsc(const GapTable &gap, uint32_t n, uint64_t seed, uint64_t first, uint64_t n_shots,
                             uint64_t *d_words, hipStream_t st);
hipError_t launch_sample_depol(const GapTable &gap, uint32_t n, uint64_t seed, uint64_t first, uint64_t n_shots,
                               uint64_t *d_x, uint64_t *d_z, hipStream_t st);
hipError_t launch_syndrome(const DeviceGraph &g, uint64_t n_shots, const uint64_t *d_words, uint64_t *d_synd,
                           hipStream_t st);
hipError_t launch_flip(const DeviceGraph &g, const FlipArgs &a, hipStream_t st);
// per-shot state in HBM (codes beyond the LDS of a CU): `big` = flip_big_scratch_bytes(g, n_shots) bytes
size_t flip_big_scratch_bytes(const DeviceGraph &g, uint64_t n_shots);
hipError_t launch_flip_big(const DeviceGraph &g, const FlipArgs &a, uint64_t *big, hipStream_t st);
hipError_t launch_flip2(const DeviceGraph &g, const NbTable &nb, const FlipArgs &a, hipStream_t st, uint32_t block_cap = 0);
// dc / dv: the common check / bit degree of a regular code, 0 when degrees vary
// groups = workgroups (64 shot slots each), waves = wavefronts per workgroup sharing a group
hipError_t launch_bp(const DeviceGraph &g, const BpArgs &a, uint32_t groups, uint32_t waves, int dc, int dv,
                     hipStream_t st);
// counters[FAILURES] += #{flags != 0}, [XFAIL] += #{flags & 1}, [ZFAIL] += #{flags & 2}
uint32_t bp_resident_groups(uint32_t waves, int dc, int dv);
size_t bp_lds_bytes(const DeviceGraph &g);
bool bp_lds_serves(const DeviceGraph &g, int dc);   // the LDS-resident BP form holds this code (a shot's messages and words within 150 KB)
bool bp_lds_hybrid_serves(const DeviceGraph &g, int dc, int dv);   // hybrid form (messages partly in LDS, partly in L2-resident global memory) can run it
bool bp_lds_hybrid_pays(const DeviceGraph &g, int dc, int dv);     // ... and measured ahead of bp_kernel (LDS holds half of a shot or more)
size_t bp_lds_workspace_bytes(const DeviceGraph &g, int dc, int dv);   // a.msg of either form
uint32_t bp_lds_workgroups(const DeviceGraph &g, int dc, int dv);   // its grid
hipError_t launch_bp_lds(const DeviceGraph &g, const BpArgs &a, int dc, int dv, hipStream_t st);
// test probe: out[i] = tanh(in[i]/2) (kind 0) or log1p(in[i]) (kind 1) with the BP kernel's own routines
hipError_t launch_bp_math_probe(int kind, uint64_t n, const double *d_in, double *d_out, hipStream_t st);
size_t bp_stream_state_bytes(uint32_t groups);
// state layout of bp_stream_run inside `state` (bp_stream_state_bytes(groups) bytes)
BpsState bp_stream_state_at(void *state, uint32_t groups);
// resume: the state already holds slots handed off by bp_kernel (no reset)
hipError_t bp_stream_run(const DeviceGraph &g, const BpArgs &a, void *state, uint32_t groups, int dc, int dv, hipStream_t st,
                         uint64_t *rounds_out, bool resume = false);
hipError_t launch_count_flags(const uint8_t *d_flags, const uint8_t *d_flags2, uint64_t n, unsigned long long *d_counters,
                              hipStream_t st);

}  // namespace ldpc
