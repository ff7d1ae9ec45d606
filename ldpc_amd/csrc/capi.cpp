// extern "C" boundary (include/ldpc_hip.h): handles, argument checking, device
// buffers and kernel launches.  No compute happens on the host here: sampling,
// syndromes, decoding and failure tests all run in the kernels of kernels.hip.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <thread>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ldpc_hip.h"
#include "host_codes.hpp"
#include "host_json.hpp"
#include "kernels.hpp"

using namespace ldpc;

// ---------------------------------------------------------------------------
// error / runtime state
// ---------------------------------------------------------------------------
namespace {
thread_local std::string g_error;
thread_local hipStream_t g_stream = nullptr;
thread_local double g_last_ms = 0.0;

int fail(int status, const std::string &msg)
{
    g_error = msg;
    return status;
}

// Nothing may unwind across the C ABI: every `int ldpc_*` entry point is a function-try-block ending in this.
#define LDPC_CATCH                                                                                  \
    catch (const std::bad_alloc &) { return fail(LDPC_ERR_CAPACITY, "out of host memory"); }       \
    catch (const std::exception &e) { return fail(LDPC_ERR_BAD_ARG, std::string("exception: ") + e.what()); } \
    catch (...) { return fail(LDPC_ERR_BAD_ARG, "unexpected exception"); }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(LDPC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
    } while (0)

struct DeviceBuffer {
    void *p = nullptr;
    size_t bytes = 0;
    ~DeviceBuffer() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t reserve(size_t want)
    {
        if (want <= bytes) return hipSuccess;
        release();
        hipError_t e = hipMalloc(&p, want ? want : 1);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    template <class T> T *as() const { return (T *)p; }
};

// kernel timing on the launch stream
struct KernelTimer {
    hipEvent_t a = nullptr, b = nullptr;
    bool ok = false;
    KernelTimer()
    {
        ok = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess;
        if (ok) (void)hipEventRecord(a, g_stream);
    }
    void stop_and_store()
    {
        if (!ok) return;
        (void)hipEventRecord(b, g_stream);
        (void)hipEventSynchronize(b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, a, b) == hipSuccess) g_last_ms = ms;
    }
    ~KernelTimer()
    {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

inline size_t words_for(size_t nbits) { return (nbits + 63) / 64; }

bool valid_probability(double p) { return p >= 0.0 && p <= 1.0; }   // noise/mod.rs:40-46

// A channel the device samplers can draw from: the gap table holds P(gap >= j) as floor(q^j 2^32) (DESIGN.md
// "Noise specification"), so below about 1e-7 the sampled rate is off by more than 0.1 % and below 2^-32 it would
// stick at 2^-32.  rand's Bernoulli (binary_symmetric_channel.rs:23-28) compares 64 bits; rather than sample a
// different channel silently, such probabilities are refused (0 itself is exact: no draws).
constexpr double MIN_SAMPLED_PROBABILITY = 1e-7;
const char *const SAMPLED_PROBABILITY_MSG =
    "probability is not 0 and below 1e-7: the 32-bit gap table of the device samplers cannot represent it to 0.1 %";
bool samplable_probability(double p) { return p == 0.0 || p >= MIN_SAMPLED_PROBABILITY; }

GapTable make_gap_table(double p)
{
    GapTable t;
    double q = 1.0 - p, qj = 1.0;
    t.q[0] = 0xFFFFFFFFu;
    for (int j = 1; j <= 64; j++) {
        qj = qj * q;
        double f = std::floor(qj * 4294967296.0);
        t.q[j] = f >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)f;
    }
    t.p_is_zero = p == 0.0;
    t.inv_log2q = (p > 0.0 && p < 1.0) ? (float)(1.0 / std::log2(1.0 - p)) : 0.0f;
    return t;
}

// strictly ascending, in range (SparseBinVec::new panics otherwise)
bool valid_positions(const size_t *pos, size_t n, size_t len)
{
    for (size_t k = 0; k < n; k++) {
        if (pos[k] >= len) return false;
        if (k && pos[k] <= pos[k - 1]) return false;
    }
    return true;
}

std::vector<uint64_t> pack(const size_t *pos, size_t n, size_t len)
{
    std::vector<uint64_t> w(std::max<size_t>(words_for(len), 1), 0);
    for (size_t k = 0; k < n; k++) w[pos[k] >> 6] |= 1ull << (pos[k] & 63);
    return w;
}

void unpack(const std::vector<uint64_t> &w, size_t len, size_t *out, size_t *out_n)
{
    size_t k = 0;
    for (size_t i = 0; i < len; i++)
        if ((w[i >> 6] >> (i & 63)) & 1) out[k++] = i;
    *out_n = k;
}
}  // namespace

// ---------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------
struct ldpc_code {
    LinearCode host;
    mutable std::mutex mu;
    mutable bool uploaded = false;
    mutable int device = -1;   // the GPU that holds the graph (and the owning decoder's buffers)
    mutable DeviceBuffer d_hptr, d_hidx, d_tptr, d_tidx, d_tedge, d_tpack, d_tpack4;
    mutable DeviceGraph dg{};
    // staging for the single-vector entry points (kept across calls: a trait-level call is one shot, and an
    // allocation and a free per call would cost more than its kernel); io_mu is held for the whole call
    mutable std::mutex io_mu;
    mutable DeviceBuffer io_in, io_out;

    int ensure_device() const
    {
        std::lock_guard<std::mutex> lock(mu);
        int current = 0;
        HIP_TRY(hipGetDevice(&current));
        if (uploaded) {
            // a kernel launched on another GPU would dereference this GPU's pointers
            if (current != device)
                return fail(LDPC_ERR_BAD_ARG, "this handle's buffers live on device " + std::to_string(device) +
                                                  ", the calling thread's current device is " + std::to_string(current) +
                                                  ": create one handle per device");
            return LDPC_OK;
        }
        const SparseBinMat &h = host.h, &t = host.ht;
        // slot of bit_adjacencies -> edge index in check-major order
        std::vector<uint32_t> tedge(t.nnz());
        {
            std::vector<uint32_t> fill(t.ptr.begin(), t.ptr.end() - 1);
            for (uint32_t r = 0; r < h.nrows(); r++)
                for (uint32_t e = h.ptr[r]; e < h.ptr[r + 1]; e++) tedge[fill[h.idx[e]]++] = e;
        }
        auto up = [](DeviceBuffer &b, const std::vector<uint32_t> &v) -> hipError_t {
            hipError_t e = b.reserve(std::max<size_t>(v.size(), 1) * 4);
            if (e != hipSuccess) return e;
            if (v.empty()) return hipSuccess;
            return hipMemcpy(b.p, v.data(), v.size() * 4, hipMemcpyHostToDevice);
        };
        HIP_TRY(up(d_hptr, h.ptr));
        HIP_TRY(up(d_hidx, h.idx));
        HIP_TRY(up(d_tptr, t.ptr));
        HIP_TRY(up(d_tidx, t.idx));
        HIP_TRY(up(d_tedge, tedge));
        dg.t_pack = nullptr;
        dg.t_pack4 = nullptr;
        if (t.nnz() && t.nnz() < 65536 && h.nrows() < 65536) {   // (small codes: bp_lds_kernel reads edge and check of a slot as one word)
            std::vector<uint32_t> pack(t.nnz());
            for (size_t k = 0; k < pack.size(); k++) pack[k] = tedge[k] | (t.idx[k] << 16);
            HIP_TRY(up(d_tpack, pack));
            dg.t_pack = d_tpack.as<uint32_t>();
            if (t.max_row_len() <= 4) {   // (bits of mixed degree up to four: a bit's words as one 16-byte row, bp_lds_kernel loads them a chunk ahead)
                std::vector<uint32_t> pack4((size_t)h.ncols * 4, 0xFFFFFFFFu);
                for (uint32_t b = 0; b < h.ncols; b++)
                    for (uint32_t k = t.ptr[b]; k < t.ptr[b + 1]; k++) pack4[(size_t)b * 4 + (k - t.ptr[b])] = pack[k];
                HIP_TRY(up(d_tpack4, pack4));
                dg.t_pack4 = d_tpack4.as<uint4>();
            }
        }
        dg.n = h.ncols;
        dg.m = h.nrows();
        dg.E = h.nnz();
        dg.max_dc = h.max_row_len();
        dg.max_dv = t.max_row_len();
        dg.h_ptr = d_hptr.as<uint32_t>();
        dg.h_idx = d_hidx.as<uint32_t>();
        dg.t_ptr = d_tptr.as<uint32_t>();
        dg.t_idx = d_tidx.as<uint32_t>();
        dg.t_edge = d_tedge.as<uint32_t>();
        device = current;
        uploaded = true;
        return LDPC_OK;
    }
};

struct ldpc_css_code {
    ldpc_code m[4];   // 0 x stabilizers, 1 z stabilizers, 2 x logicals, 3 z logicals
};

struct ldpc_flip_decoder {
    ldpc_code code;
    std::mutex mu;
    DeviceBuffer counters, nb, o_scratch, flip_log;
    std::mutex io_mu;               // single-vector staging, as in ldpc_code
    DeviceBuffer io_in, io_out;
    NbTable nbt{};
    Nb3Table nbt3{};
    // 0 = undecided, 1 = CSR kernel (one wave per shot), 2 = neighbourhood-table kernel (flip2_kernel),
    // 3 = eight shots per wave (flip3_kernel: regular codes of bit degree 3, n < 16384)
    int path = 0;
};

struct ldpc_bp_decoder {
    ldpc_code code;
    double p = 0.5;
    uint32_t max_iter = 0;
    std::mutex mu;
    DeviceBuffer counters, workspace, stream_state;
    std::mutex io_mu;               // single-vector staging, as in ldpc_code
    DeviceBuffer io_in, io_out;
    int reg_dc = 0, reg_dv = 0;   // common check / bit degree when the code is regular, else 0 (reg_dv = -4: mixed, none above 4)
};

struct ldpc_css_bp_decoder {
    ldpc_bp_decoder x;   // BpDecoder on Hx: decodes syndrome.x, corrects the z part
    ldpc_bp_decoder z;   // BpDecoder on Hz: decodes syndrome.z, corrects the x part
    ldpc_code lx, lz;
    std::mutex mu;
    DeviceBuffer flags, flags2, counters;
    hipStream_t side_stream = nullptr;   // for the half that runs beside the caller's stream (single-call entry point)
    ~ldpc_css_bp_decoder()
    {
        if (side_stream) (void)hipStreamDestroy(side_stream);
    }
};

namespace {

void copy_code(ldpc_code &dst, const LinearCode &src) { dst.host = src; }

ldpc_code *new_code(LinearCode c)
{
    ldpc_code *h = new ldpc_code();
    h->host = std::move(c);
    return h;
}

void export_csr(const SparseBinMat &m, size_t *row_ptr, size_t *col_idx)
{
    for (size_t i = 0; i < m.ptr.size(); i++) row_ptr[i] = m.ptr[i];
    for (size_t i = 0; i < m.idx.size(); i++) col_idx[i] = m.idx[i];
}

// iterations the calling thread's last BP call actually ran (ldpc_last_bp_executed_iterations)
thread_local uint64_t g_bp_executed = 0;

// the counters of a handle: LDPC_COUNTERS of them, then the work-queue word, the hand-off flag (bp_kernel) and the
// count of iterations actually run (bp_lds_kernel)
constexpr int COUNTER_WORDS = LDPC_COUNTERS + 3;

int read_counters(DeviceBuffer &d, uint64_t *out, uint64_t *executed = nullptr)
{
    unsigned long long tmp[COUNTER_WORDS];
    HIP_TRY(hipMemcpyAsync(tmp, d.p, sizeof(tmp), hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    for (int i = 0; i < LDPC_COUNTERS; i++) out[i] = tmp[i];
    if (executed) *executed = tmp[LDPC_COUNTERS + 2];
    return LDPC_OK;
}

int zero_counters(DeviceBuffer &d)
{
    HIP_TRY(d.reserve(sizeof(unsigned long long) * COUNTER_WORDS));
    HIP_TRY(hipMemsetAsync(d.p, 0, sizeof(unsigned long long) * COUNTER_WORDS, g_stream));
    return LDPC_OK;
}

// ---------------------------------------------------------------------------
// flip driver
// ---------------------------------------------------------------------------
// Neighbourhood table of the fast flip kernel (kernels.hpp NbTable).  Returns
// false when the code does not suit it (bit degree > 7, table > 1 GiB).
bool build_nb_table(const LinearCode &c, std::vector<uint32_t> &out, uint32_t &kshift, uint32_t &dvp)
{
    const SparseBinMat &h = c.h, &t = c.ht;
    const uint32_t n = h.ncols;
    const uint32_t max_dv = t.max_row_len();
    if (max_dv > 7 || h.nrows() >= (1u << 19) - 64) return false;   // packed check format: word offsets below 2^16
    dvp = max_dv <= 3 ? 3 : 7;
    std::vector<uint32_t> stamp(n, 0xFFFFFFFFu);
    uint32_t kmax = 1;
    for (uint32_t b = 0; b < n; b++) {
        uint32_t cnt = 1;
        stamp[b] = b;
        for (uint32_t k = t.ptr[b]; k < t.ptr[b + 1]; k++)
            for (uint32_t e = h.ptr[t.idx[k]]; e < h.ptr[t.idx[k] + 1]; e++)
                if (stamp[h.idx[e]] != b) { stamp[h.idx[e]] = b; cnt++; }
        kmax = std::max(kmax, cnt);
    }
    kshift = 4;
    while ((1u << kshift) < kmax) kshift++;
    const size_t K = (size_t)1 << kshift, rec = 1 + dvp;
    if ((size_t)n * K * rec * 4 > ((size_t)1 << 30)) return false;
    out.assign((size_t)n * K * rec, 0xFFFFFFFFu);
    std::fill(stamp.begin(), stamp.end(), 0xFFFFFFFFu);
    auto put = [&](uint32_t b, size_t slot, uint32_t bit) {
        uint32_t *r = &out[((size_t)b * K + slot) * rec];
        r[0] = bit;
        // a check c is stored as (byte offset of its 32-bit syndrome word) << 16 | bit in that word
        for (uint32_t k = t.ptr[bit]; k < t.ptr[bit + 1]; k++) r[1 + (k - t.ptr[bit])] = ((t.idx[k] >> 5) * 4u) << 16 | (t.idx[k] & 31u);
    };
    for (uint32_t b = 0; b < n; b++) {
        size_t slot = 0;
        stamp[b] = b;
        put(b, slot++, b);
        for (uint32_t k = t.ptr[b]; k < t.ptr[b + 1]; k++)
            for (uint32_t e = h.ptr[t.idx[k]]; e < h.ptr[t.idx[k] + 1]; e++) {
                uint32_t bb = h.idx[e];
                if (stamp[bb] != b) { stamp[bb] = b; put(b, slot++, bb); }
            }
    }
    return true;
}

// Table of flip3_kernel (kernels.hpp Nb3Table).  Returns false when the code does not suit it.
bool build_nb3_table(const LinearCode &c, std::vector<uint32_t> &out, Nb3Table &tab)
{
    const SparseBinMat &h = c.h, &t = c.ht;
    const uint32_t n = h.ncols, m = h.nrows();
    if (n == 0 || m == 0 || n + 512 > 32 * 512) return false;
    for (uint32_t b = 0; b < n; b++)
        if (t.row_len(b) != 3) return false;
    tab.sw = (m + 64) / 64;
    tab.windows = (n + 512) / 512;
    const uint32_t spare_bit = tab.windows * 512 - 1, spare_chk = m;
    auto pack = [](uint32_t i) { return ((i >> 5) * 4u) << 16 | (i & 31u); };
    out.assign((size_t)(n + 1) * 16 * 4, 0);
    for (size_t s = 0; s < (size_t)(n + 1) * 16; s++) {
        out[s * 4] = pack(spare_bit);
        out[s * 4 + 1] = out[s * 4 + 2] = out[s * 4 + 3] = pack(spare_chk);
    }
    // Record of a slot: word 0 = mask position of the slot's bit | flags, words 1, 2 = the checks the flip loop reads
    // from the syndrome, word 3 = the check that connects the slot's bit to the row's bit b (the loop takes its new
    // value from the lane that toggled it).  Flags: bit 8 = "always rewrite the mask bit" (b itself, and a bit that
    // shares more than one check with b: its old state cannot be told from one check), bits 9-10 = which of b's three
    // checks word 3 is.  Slot 0 is b itself (word 3 = its third check).
    std::vector<uint32_t> stamp(n, 0xFFFFFFFFu);
    for (uint32_t b = 0; b < n; b++) {
        size_t slot = 0;
        auto put = [&](uint32_t bit, uint32_t conn_k, bool always) {
            uint32_t *r = &out[((size_t)b * 16 + slot++) * 4];
            const uint32_t conn = t.idx[t.ptr[b] + conn_k];
            r[0] = pack(bit) | (always ? 1u << 8 : 0u) | conn_k << 9;
            uint32_t w = 1;
            bool seen = false;
            for (uint32_t k = 0; k < 3; k++) {
                const uint32_t c = t.idx[t.ptr[bit] + k];
                if (c == conn && !seen) seen = true;
                else r[w++] = pack(c);
            }
            r[3] = pack(conn);
        };
        stamp[b] = b;
        put(b, 2, true);
        for (uint32_t k = t.ptr[b]; k < t.ptr[b + 1]; k++)
            for (uint32_t e = h.ptr[t.idx[k]]; e < h.ptr[t.idx[k] + 1]; e++) {
                const uint32_t bb = h.idx[e];
                if (stamp[bb] == b) continue;
                if (slot == 16) return false;   // more than 15 neighbours
                stamp[bb] = b;
                uint32_t shared = 0;
                for (uint32_t k2 = 0; k2 < 3; k2++)
                    for (uint32_t k3 = 0; k3 < 3; k3++) shared += t.idx[t.ptr[bb] + k2] == t.idx[t.ptr[b] + k3];
                put(bb, k - t.ptr[b], shared > 1);
            }
    }
    return true;
}

int run_flip(ldpc_flip_decoder *dec, FlipArgs a, uint64_t *counters_out)
{
    int st = dec->code.ensure_device();
    if (st) return st;
    std::lock_guard<std::mutex> lock(dec->mu);
    st = zero_counters(dec->counters);
    if (st) return st;
    a.counters = dec->counters.as<unsigned long long>();
    a.max_flips = switches().flip_maxflips;   // profiling aid
    const DeviceGraph &g = dec->code.dg;
    // (the choice of kernel is made once per handle; LDPC_HIP_FLIP_KERNEL = "csr" / "v2" forces the older forms: tests)
    if (dec->path == 0) {
        dec->path = 1;
        {   // (this GPU's address of the static scratch array is looked up here, once, so that ldpc_flip_mc_run_noalloc
            // itself makes no call that is not a stream operation)
            unsigned char *p_ = nullptr;
            size_t b_ = 0;
            HIP_TRY(noalloc_scratch(&p_, &b_));
        }
        std::vector<uint32_t> tab;
        uint32_t kshift = 0, dvp = 0;
        const bool allow_fast = switches().flip_kernel != 1;
        Nb3Table t3{};
        if (allow_fast && switches().flip_kernel != 2 && build_nb3_table(dec->code.host, tab, t3) &&
            flip3_lds_bytes(t3, 1024) <= 64 * 1024) {
            HIP_TRY(dec->nb.reserve(tab.size() * 4));
            HIP_TRY(hipMemcpy(dec->nb.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
            t3.rec = dec->nb.as<uint4>();
            dec->nbt3 = t3;
            dec->path = 3;
        } else if (allow_fast && flip2_lds_bytes(g, 16, 1024) <= 64 * 1024 && g.n < (1u << 26) &&
            build_nb_table(dec->code.host, tab, kshift, dvp)) {
            HIP_TRY(dec->nb.reserve(std::max<size_t>(tab.size(), 1) * 4));
            if (!tab.empty()) HIP_TRY(hipMemcpy(dec->nb.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
            dec->nbt.rec = dec->nb.as<uint32_t>();
            dec->nbt.kshift = kshift;
            dec->nbt.dvp = dvp;
            {
                const SparseBinMat &t = dec->code.host.ht;
                bool reg = true;
                for (uint32_t b2 = 0; b2 < t.nrows(); b2++) reg = reg && t.row_len(b2) == dvp;
                dec->nbt.regular = reg ? 1u : 0u;
            }
            dec->path = 2;
        }
    }
    if (dec->path == 2 || dec->path == 3) {
        if (dec->path == 3) a.list_cap = a.list_cap ? flip3_list_cap(a.expected_weight) : 512;   // given words: weight unknown
        else if (!a.list_cap) a.list_cap = 512;
        // large runs of the third form take the fourth (one workgroup per CU, flip waves fed by service waves);
        // small ones would leave most of its slots empty
        const bool v4 = dec->path == 3 && switches().flip_kernel != 3 &&
                        a.n_shots >= (uint64_t)device_cu_count() * 256 && flip4_slots(dec->nbt3, a.list_cap, 4) >= 33;
        // HBM scratch pieces: one parked word and one flip log per shot in flight
        const size_t pieces = v4 ? flip4_scratch_pieces() : (size_t)(dec->path == 3 ? flip3_max_blocks() : flip2_max_blocks(g)) * 8;
        if (!a.errors_out) {
            HIP_TRY(dec->o_scratch.reserve(pieces * std::max<size_t>(words_for(g.n), 1) * 8));
            a.o_scratch = dec->o_scratch.as<uint64_t>();
        }
        a.flip_log_cap = ((g.m + 15u) & ~15u) + 16u;
        HIP_TRY(dec->flip_log.reserve(pieces * a.flip_log_cap * 4));
        a.flip_log = dec->flip_log.as<uint32_t>();
        KernelTimer timer;
        if (v4) HIP_TRY(launch_flip4(g, dec->nbt3, a, g_stream));
        else if (dec->path == 3) HIP_TRY(launch_flip3(g, dec->nbt3, a, switches().flip_prof, g_stream));
        else HIP_TRY(launch_flip2(g, dec->nbt, a, g_stream));
        timer.stop_and_store();
        if (v4) {
            // the kernel's waves give up rather than wait for ever (a bug, not a run-time condition): never report
            // counters of a run that did not finish
            uint64_t c[LDPC_COUNTERS];
            int rc = read_counters(dec->counters, c);
            if (rc) return rc;
            if (c[LDPC_CNT_SHOTS] != a.n_shots) return fail(LDPC_ERR_HIP, "flip decoder: the kernel stopped early (watchdog)");
            if (counters_out) std::memcpy(counters_out, c, sizeof(c));
            return LDPC_OK;
        }
        if (counters_out) return read_counters(dec->counters, counters_out);
        HIP_TRY(hipStreamSynchronize(g_stream));
        return LDPC_OK;
    }
    size_t per_wave = (words_for(g.m) + 2 * words_for(g.n)) * 8;
    if (33 * 8 + per_wave > 160 * 1024) {
        // the state of a shot does not fit the LDS of a CU (n = 10^6): wave-per-shot kernel with the state in HBM
        HIP_TRY(dec->o_scratch.reserve(flip_big_scratch_bytes(g, a.n_shots)));
        KernelTimer timer;
        HIP_TRY(launch_flip_big(g, a, dec->o_scratch.as<uint64_t>(), g_stream));
        timer.stop_and_store();
        if (counters_out) return read_counters(dec->counters, counters_out);
        HIP_TRY(hipStreamSynchronize(g_stream));
        return LDPC_OK;
    }
    KernelTimer timer;
    HIP_TRY(launch_flip(g, a, g_stream));
    timer.stop_and_store();
    if (counters_out) return read_counters(dec->counters, counters_out);
    HIP_TRY(hipStreamSynchronize(g_stream));
    return LDPC_OK;
}

// ---------------------------------------------------------------------------
// BP driver
// ---------------------------------------------------------------------------
struct BpRun {
    int source = SRC_BSC, test = TEST_NONE, xor_input = 0;
    uint32_t fail_bit = 1;
    GapTable gap{};
    uint64_t seed = 0, first_shot = 0, n_shots = 0;
    const uint64_t *words_in = nullptr, *synd_in = nullptr;
    uint64_t *corr_out = nullptr;
    uint32_t *iters_out = nullptr;
    double *post_out = nullptr, *chk_out = nullptr;
    uint8_t *failed_out = nullptr;
    const ldpc_code *logicals = nullptr;
    uint32_t share = 1;   // this run has 1 / share of the GPU's workgroup slots (CSS halves side by side)
};

int run_bp(ldpc_bp_decoder *dec, const BpRun &r, uint64_t *counters_out, bool time_it = true)
{
    int st = dec->code.ensure_device();
    if (st) return st;
    if (r.logicals && (st = r.logicals->ensure_device())) return st;
    std::lock_guard<std::mutex> lock(dec->mu);
    st = zero_counters(dec->counters);
    if (st) return st;
    const DeviceGraph &g = dec->code.dg;
    g_bp_executed = 0;   // (an empty call reports no iterations, not the previous call's)
    if (r.n_shots == 0) {
        if (counters_out) return read_counters(dec->counters, counters_out);
        return LDPC_OK;
    }
    // shot groups of 64 slots, one workgroup each.  Waves per group grow with the code (an iteration of
    // a 64-shot group costs ~E row updates: split over up to 8 waves so that a straggler shot does not
    // hold a long iteration on one wave); groups are bounded by memory and by what is resident at once.
    size_t per_group = bp_workspace_bytes_per_group(g);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    size_t budget = std::min<size_t>((size_t)((double)(free_b + dec->workspace.bytes) * 0.75), (size_t)200 << 30);
    // measured (scripts/bp_probe.py, scripts/css_one.py): C2 (E = 30000), C1 (E = 3600) and the 625-qubit HGP code
    // (E = 2100 per half) all run fastest with 8 waves per group: shorter iterations, shorter straggler tail
    const uint32_t waves_default = g.E <= 256 ? 1u : g.E <= 1024 ? 2u : 8u;
    uint32_t waves = waves_default;
    if (switches().bp_waves) waves = (uint32_t)switches().bp_waves;   // A/B aid
    // only as many groups as are resident at once: a queued workgroup would find the shot queue drained
    const uint64_t over = (uint64_t)switches().bp_oversub;   // A/B aid
    uint64_t max_groups = std::min<uint64_t>(std::max<uint64_t>(over * bp_resident_groups(waves, dec->reg_dc, dec->reg_dv) / r.share, 1),
                                             budget / std::max<size_t>(per_group, 1) / r.share);
    if (max_groups == 0)
        return fail(LDPC_ERR_CAPACITY, "bp decoder: one shot group needs " + std::to_string(per_group) +
                                           " bytes of message storage, more than the device can give");
    uint64_t groups = std::min<uint64_t>(max_groups, (r.n_shots + 63) / 64);
    groups = std::max<uint64_t>(groups, 1);
    // Streamed form (one grid-wide kernel per phase, all CUs on every group) when memory, not the shot count,
    // leaves the persistent kernel with fewer workgroups than half the CUs: C5, 3 GB of messages per group.
    // ... and when the call itself is small: up to 256 groups of work (one trait-level call is a single shot) leave
    // the persistent kernel one workgroup per group and most CUs idle, each iteration paying the latency of a whole
    // pass over the graph; the streamed kernels put every CU on those groups.  Measured (scripts/bp_small_batch.py):
    // one shot at C2 10.5 -> 0.34 ms, 64 groups 17 -> 4.8 ms, even at 512 groups; between E = 200 and 3000 the
    // streamed form still wins up to 128 groups (E = 2100: one shot 0.65 -> 0.28 ms), below E = 200 a round's
    // launches cost more than the persistent iteration.
    const uint64_t by_memory = budget / std::max<size_t>(per_group, 1);
    const uint64_t work_groups = (r.n_shots + 63) / 64;
    bool stream = by_memory < 128 || (g.E >= 3000 && work_groups <= 256) || (g.E >= 200 && work_groups <= 128);
    if (switches().bp_path == 1) stream = true;   // A/B and test aid
    else if (switches().bp_path == 2) stream = false;
    // Codes whose messages of a shot fit LDS (E * 8 B + words <= 150 KB): one workgroup per shot in flight
    // (bp_lds_kernel), of 256 threads where five or more shots fit a CU and up to 1024 where fewer do (bp_lds_shape).
    // Measured against the forms above: one shot 0.3 -> 0.07 ms; 2^22 shots C1 2.0x, CSS example 4x; codes of which a CU
    // holds one shot, 2^20 shots: (3,6) E = 15,000 1.6x, E = 18,000 1.4-1.6x, (3,7) E = 12,600 1.8x (scripts/bp_lds_big.py;
    // with 256-thread workgroups these lost, 0.83-0.9x, and the rule used to send them to bp_kernel).  Posterior /
    // message outputs stay with the forms that keep both message arrays.  LDPC_HIP_BP_PATH = "lds" / "hbm" force and
    // forbid it.
    const bool lds_form = !r.post_out && !r.chk_out && ((bp_lds_serves(g, dec->reg_dc) && (switches().bp_path == 3 || switches().bp_path == 0)) ||
                                                        (bp_lds_hybrid_serves(g, dec->reg_dc, dec->reg_dv) && (switches().bp_path == 3 || (switches().bp_path == 0 && bp_lds_hybrid_pays(g, dec->reg_dc, dec->reg_dv)))));
    if (lds_form) {
        stream = false;
        groups = 0;
    }
    if (stream) groups = std::min<uint64_t>(std::min<uint64_t>(by_memory, 4096), std::max<uint64_t>((r.n_shots + 63) / 64, 1));
    if (switches().bp_groups && stream)   // A/B aid (streamed form only: the persistent grid must stay resident)
        groups = std::max<uint64_t>(1, std::min<uint64_t>(groups, (uint64_t)switches().bp_groups));
    if (switches().debug)
        std::fprintf(stderr, "[ldpc_hip] bp: %s E=%u waves=%u groups=%llu resident=%u per_group=%zu MiB\n", lds_form ? "lds" : stream ? "stream" : "persistent", g.E, waves,
                     (unsigned long long)groups, bp_resident_groups(waves, dec->reg_dc, dec->reg_dv), per_group >> 20);
    if (lds_form) HIP_TRY(dec->workspace.reserve(bp_lds_workspace_bytes(g, dec->reg_dc, dec->reg_dv)));   // (saved states, bp_lds_kernel)
    else HIP_TRY(dec->workspace.reserve(per_group * groups));

    BpArgs a{};
    a.source = r.source;
    a.test = r.test;
    a.xor_input = r.xor_input;
    a.max_iter = dec->max_iter;
    a.fail_bit = r.fail_bit;
    a.profile = switches().bp_profile;
    a.prior = std::log((1.0 - dec->p) / dec->p);   // belief_propagation.rs:38
    a.seed = r.seed;
    a.first_shot = r.first_shot;
    a.n_shots = r.n_shots;
    a.words_in = r.words_in;
    a.synd_in = r.synd_in;
    a.corr_out = r.corr_out;
    a.iters_out = r.iters_out;
    a.post_out = r.post_out;
    a.chk_out = r.chk_out;
    a.failed_out = r.failed_out;
    a.counters = dec->counters.as<unsigned long long>();
    a.queue = dec->counters.as<unsigned long long>() + LDPC_COUNTERS;   // zeroed with the counters
    a.executed = dec->counters.as<unsigned long long>() + LDPC_COUNTERS + 2;
    if (r.logicals) {
        a.l_rows = r.logicals->dg.m;
        a.l_nnz = r.logicals->dg.E;
        a.l_ptr = r.logicals->dg.h_ptr;
        a.l_idx = r.logicals->dg.h_idx;
    }
    size_t Wn = words_for(g.n);
    char *base = dec->workspace.as<char>();
    a.msg = (double *)base;
    base += groups * 2 * (size_t)g.E * 64 * sizeof(double);
    a.ew = (uint64_t *)base;
    base += groups * Wn * 64 * 8;
    a.dw = (uint64_t *)base;
    base += groups * Wn * 64 * 8;
    a.dsl = (uint64_t *)base;
    base += groups * (size_t)g.n * 8;
    a.ssl = (uint64_t *)base;
    base += groups * (size_t)g.m * 8;
    a.esl = (uint64_t *)base;
    base += groups * (size_t)g.n * 8;
    a.psl = (uint64_t *)base;
    a.gap = r.gap;

    std::unique_ptr<KernelTimer> timer(time_it ? new KernelTimer() : nullptr);
    if (lds_form) {
        HIP_TRY(launch_bp_lds(g, a, dec->reg_dc, dec->reg_dv, g_stream));
    } else if (stream) {
        HIP_TRY(dec->stream_state.reserve(bp_stream_state_bytes((uint32_t)groups)));
        uint64_t rounds = 0;
        HIP_TRY(bp_stream_run(g, a, dec->stream_state.as<char>(), (uint32_t)groups, dec->reg_dc, dec->reg_dv, g_stream, &rounds));
        if (switches().debug) std::fprintf(stderr, "[ldpc_hip] bp stream: %llu rounds\n", (unsigned long long)rounds);
    } else {
        // tail hand-off (kernels.hpp, BpArgs::handoff_iter): shots still running 16 iterations after the queue
        // has drained are finished by the streamed form, the whole chip on the few groups that hold them
        uint32_t handoff = switches().bp_handoff;   // LDPC_HIP_BP_HANDOFF: 0 = off
        // (the streamed tail packs the stragglers of all groups into few (bps_compact), which the persistent
        // kernel cannot do, and a streamed iteration of a small tail costs about 0.05 ms of launches: below
        // E = 200 a lone group's persistent iteration is cheaper than that; scripts/bp_small_batch.py)
        const bool forced = switches().bp_handoff_set;
        if (handoff == 0 || dec->max_iter <= handoff || (!forced && g.E < 200)) handoff = 0xFFFFFFFFu;
        a.handoff_iter = handoff;
        if (handoff != 0xFFFFFFFFu) {
            HIP_TRY(dec->stream_state.reserve(bp_stream_state_bytes((uint32_t)groups)));
            HIP_TRY(hipMemsetAsync(dec->stream_state.p, 0, bp_stream_state_bytes((uint32_t)groups), g_stream));
            a.handoff = bp_stream_state_at(dec->stream_state.p, (uint32_t)groups);
            a.handoff_flag = dec->counters.as<unsigned long long>() + LDPC_COUNTERS + 1;
        }
        const auto t0 = std::chrono::steady_clock::now();
        HIP_TRY(launch_bp(g, a, (uint32_t)groups, waves, dec->reg_dc, dec->reg_dv, g_stream));
        if (handoff != 0xFFFFFFFFu) {
            unsigned long long flag = 0;
            HIP_TRY(hipMemcpyAsync(&flag, a.handoff_flag, 8, hipMemcpyDeviceToHost, g_stream));
            HIP_TRY(hipStreamSynchronize(g_stream));
            if (flag) {
                const auto t1 = std::chrono::steady_clock::now();
                uint64_t rounds = 0;
                HIP_TRY(bp_stream_run(g, a, dec->stream_state.p, (uint32_t)groups, dec->reg_dc, dec->reg_dv, g_stream, &rounds, true));
                if (switches().debug)
                    std::fprintf(stderr, "[ldpc_hip] bp tail hand-off: persistent kernel %.2f ms, then %llu streamed rounds in %.2f ms\n",
                                 std::chrono::duration<double, std::milli>(t1 - t0).count(), (unsigned long long)rounds,
                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
            }
        }
    }
    if (timer) timer->stop_and_store();
    g_bp_executed = 0;
    if (counters_out) {
        uint64_t ran = 0;
        const int rc = read_counters(dec->counters, counters_out, &ran);
        // (only the LDS form can run fewer iterations than it reports)
        g_bp_executed = lds_form ? ran : counters_out[LDPC_CNT_WORK];
        return rc;
    }
    HIP_TRY(hipStreamSynchronize(g_stream));
    return LDPC_OK;
}

int init_bp(ldpc_bp_decoder *d, const LinearCode &code, double p, size_t max_iterations)
{
    if (!valid_probability(p)) return fail(LDPC_ERR_BAD_ARG, "probability is not between 0 and 1");
    if (max_iterations > 0xFFFFFFFFull) return fail(LDPC_ERR_BAD_ARG, "max_iterations too large");
    copy_code(d->code, code);
    {
        const SparseBinMat &h = d->code.host.h, &t = d->code.host.ht;
        int dc = h.nrows() ? (int)h.row_len(0) : 0, dv = t.nrows() ? (int)t.row_len(0) : 0;
        for (uint32_t r = 0; r < h.nrows(); r++)
            if ((int)h.row_len(r) != dc) dc = 0;
        for (uint32_t r = 0; r < t.nrows(); r++)
            if ((int)t.row_len(r) != dv) dv = 0;
        // (bits of mixed degree, none above 4 -- the hypergraph-product codes: the any-degree bit update with row
        // registers for four checks instead of eight)
        if (dv == 0 && t.nrows() && t.max_row_len() <= 4) dv = -4;
        d->reg_dc = dc;
        d->reg_dv = dv;
    }
    d->p = p;
    d->max_iter = (uint32_t)max_iterations;
    return LDPC_OK;
}

}  // namespace

// ===========================================================================
// extern "C"
// ===========================================================================
extern "C" {

const char *ldpc_last_error(void) { return g_error.c_str(); }

int ldpc_device_count(int *count)
try {
    if (!count) return fail(LDPC_ERR_BAD_ARG, "null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return fail(LDPC_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = c;
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_set_device(int device)
try {
    HIP_TRY(hipSetDevice(device));
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_set_stream(void *hip_stream)
try {
    g_stream = (hipStream_t)hip_stream;
    return LDPC_OK;
}
LDPC_CATCH

double ldpc_last_kernel_ms(void) { return g_last_ms; }

void ldpc_debug_reload_switches(void) { reload_switches(); }

const char *ldpc_last_kernel_name(void) { return last_launched_kernel(); }
uint64_t ldpc_last_bp_executed_iterations(void) { return g_bp_executed; }

// ---- LinearCode -----------------------------------------------------------
int ldpc_code_create(size_t n_bits, size_t n_checks, const size_t *row_ptr, const size_t *col_idx, ldpc_code **out)
try {
    if (!out || (!row_ptr && n_checks)) return fail(LDPC_ERR_BAD_ARG, "null argument");
    if (n_checks && row_ptr[n_checks] > 0 && !col_idx) return fail(LDPC_ERR_BAD_ARG, "null argument");
    if (n_bits >= 0xFFFFFFFFull || n_checks >= 0xFFFFFFFFull) return fail(LDPC_ERR_CAPACITY, "dimension exceeds 32 bits");
    std::vector<std::vector<uint32_t>> rows(n_checks);
    if (n_checks) {
        if (row_ptr[0] != 0) return fail(LDPC_ERR_BAD_ARG, "row_ptr[0] must be 0");
        if (row_ptr[n_checks] >= 0xFFFFFFFFull) return fail(LDPC_ERR_CAPACITY, "too many edges");
        for (size_t r = 0; r < n_checks; r++) {
            if (row_ptr[r + 1] < row_ptr[r]) return fail(LDPC_ERR_BAD_ARG, "row_ptr must be non-decreasing");
            for (size_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
                if (col_idx[e] >= n_bits) return fail(LDPC_ERR_BAD_ARG, "column index out of bound");
                rows[r].push_back((uint32_t)col_idx[e]);
            }
        }
    }
    SparseBinMat h;
    std::string err;
    if (!SparseBinMat::from_rows((uint32_t)n_bits, rows, h, err)) return fail(LDPC_ERR_BAD_ARG, err);
    *out = new_code(make_code(std::move(h)));
    return LDPC_OK;
}
LDPC_CATCH

// rows (row_ptr, col_idx) -> SparseBinMat with the checks of ldpc_code_create; 0 on success
static int matrix_from_csr(size_t n_cols, size_t n_rows, const size_t *row_ptr, const size_t *col_idx, SparseBinMat &out)
{
    if (!row_ptr && n_rows) return fail(LDPC_ERR_BAD_ARG, "null argument");
    if (n_cols >= 0xFFFFFFFFull || n_rows >= 0xFFFFFFFFull) return fail(LDPC_ERR_CAPACITY, "dimension exceeds 32 bits");
    std::vector<std::vector<uint32_t>> rows(n_rows);
    if (n_rows) {
        if (row_ptr[0] != 0) return fail(LDPC_ERR_BAD_ARG, "row_ptr[0] must be 0");
        if (row_ptr[n_rows] >= 0xFFFFFFFFull) return fail(LDPC_ERR_CAPACITY, "too many edges");
        if (row_ptr[n_rows] && !col_idx) return fail(LDPC_ERR_BAD_ARG, "null argument");
        for (size_t r = 0; r < n_rows; r++) {
            if (row_ptr[r + 1] < row_ptr[r]) return fail(LDPC_ERR_BAD_ARG, "row_ptr must be non-decreasing");
            for (size_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
                if (col_idx[e] >= n_cols) return fail(LDPC_ERR_BAD_ARG, "column index out of bound");
                rows[r].push_back((uint32_t)col_idx[e]);
            }
        }
    }
    std::string err;
    if (!SparseBinMat::from_rows((uint32_t)n_cols, rows, out, err)) return fail(LDPC_ERR_BAD_ARG, err);
    return LDPC_OK;
}

int ldpc_code_from_generator_matrix(size_t n_bits, size_t n_generators, const size_t *row_ptr, const size_t *col_idx, ldpc_code **out)
try {
    if (!out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    SparseBinMat gmat;
    int st = matrix_from_csr(n_bits, n_generators, row_ptr, col_idx, gmat);
    if (st) return st;
    *out = new_code(make_code(nullspace(gmat)));   // linear_code/mod.rs:116-124
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_code_from_both_matrices(size_t n_bits, size_t n_generators, const size_t *g_row_ptr, const size_t *g_col_idx, size_t n_checks,
                                 const size_t *h_row_ptr, const size_t *h_col_idx, ldpc_code **out)
try {
    if (!out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    SparseBinMat gmat, h;
    int st = matrix_from_csr(n_bits, n_generators, g_row_ptr, g_col_idx, gmat);
    if (st) return st;
    if ((st = matrix_from_csr(n_bits, n_checks, h_row_ptr, h_col_idx, h))) return st;
    // linear_code/mod.rs:59-72 (both matrices are given n_bits columns here, so only the second panic can occur)
    if (!h.product_is_zero(gmat)) return fail(LDPC_ERR_BAD_ARG, "matrices are non orthogonal");
    *out = new_code(make_code(std::move(h)));
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_code_hamming(ldpc_code **out)
try {
    if (!out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    *out = new_code(hamming_code());
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_code_repetition(size_t length, ldpc_code **out)
try {
    if (!out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    if (length == 0 || length >= 0xFFFFFFFFull) return fail(LDPC_ERR_BAD_ARG, "invalid repetition code length");
    *out = new_code(repetition_code((uint32_t)length));
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_code_random_regular(size_t n_bits, size_t n_checks, size_t bit_degree, size_t check_degree, uint64_t seed,
                             ldpc_code **out)
try {
    if (!out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    LinearCode c;
    if (n_bits >= 0xFFFFFFFFull || n_checks >= 0xFFFFFFFFull || bit_degree >= 0xFFFFFFFFull ||
        check_degree >= 0xFFFFFFFFull ||
        !random_regular_code((uint32_t)n_bits, (uint32_t)n_checks, (uint32_t)bit_degree, (uint32_t)check_degree, seed, c))
        return fail(LDPC_ERR_SAMPLING, "can't generate a regular code with " + std::to_string(n_bits) +
                                           " bits of degree " + std::to_string(bit_degree) + " and " +
                                           std::to_string(n_checks) + " checks of degree " +
                                           std::to_string(check_degree));   // random.rs:97-104
    *out = new_code(std::move(c));
    return LDPC_OK;
}
LDPC_CATCH

size_t ldpc_code_len(const ldpc_code *code) { return code ? code->host.len() : 0; }
size_t ldpc_code_num_checks(const ldpc_code *code) { return code ? code->host.num_checks() : 0; }
size_t ldpc_code_num_edges(const ldpc_code *code) { return code ? code->host.h.nnz() : 0; }

int ldpc_code_parity_check_matrix(const ldpc_code *code, size_t *row_ptr, size_t *col_idx)
try {
    if (!code || !row_ptr || (!col_idx && code->host.h.nnz())) return fail(LDPC_ERR_BAD_ARG, "null argument");
    export_csr(code->host.h, row_ptr, col_idx);
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_code_bit_adjacencies(const ldpc_code *code, size_t *row_ptr, size_t *col_idx)
try {
    if (!code || !row_ptr || (!col_idx && code->host.h.nnz())) return fail(LDPC_ERR_BAD_ARG, "null argument");
    export_csr(code->host.ht, row_ptr, col_idx);
    return LDPC_OK;
}
LDPC_CATCH

// ---- host algebra (small codes) ---------------------------------------------
int ldpc_code_num_generators(const ldpc_code *code, size_t *out)
try {
    if (!code || !out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    *out = code->host.len() - gf2_rank(code->host.h);   // rows of a nullspace basis
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_code_generator_matrix(const ldpc_code *code, size_t *row_ptr, size_t *col_idx, size_t col_idx_capacity)
try {
    if (!code || !row_ptr) return fail(LDPC_ERR_BAD_ARG, "null argument");
    const SparseBinMat gmat = nullspace(code->host.h);
    if (gmat.nnz() > col_idx_capacity || (!col_idx && gmat.nnz()))
        return fail(LDPC_ERR_CAPACITY, "generator matrix has " + std::to_string(gmat.nnz()) + " ones, buffer holds " +
                                           std::to_string(col_idx_capacity));
    export_csr(gmat, row_ptr, col_idx);
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_code_dimension(const ldpc_code *code, size_t *out)
try {
    // generator_matrix.rank() (mod.rs:308-310); the rows of a nullspace basis are independent
    return ldpc_code_num_generators(code, out);
}
LDPC_CATCH

int ldpc_code_minimal_distance(const ldpc_code *code, size_t *distance, int *found)
try {
    if (!code || !distance || !found) return fail(LDPC_ERR_BAD_ARG, "null argument");
    uint32_t w = 0;
    if (!minimal_row_span_weight(nullspace(code->host.h), w))
        return fail(LDPC_ERR_CAPACITY, "minimal_distance enumerates 2^dimension codewords: dimension above 30");
    *found = w != 0;
    *distance = w;
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_code_has_same_codespace(const ldpc_code *code, const ldpc_code *other, int *out)
try {
    if (!code || !other || !out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    // mod.rs:272-275: same length and H * G_other^T == 0
    *out = code->host.len() == other->host.len() && code->host.h.product_is_zero(nullspace(other->host.h));
    return LDPC_OK;
}
LDPC_CATCH

static int give_text(const std::string &s, char *buf, size_t capacity, size_t *needed)
{
    *needed = s.size() + 1;
    if (capacity < s.size() + 1 || !buf)
        return fail(LDPC_ERR_CAPACITY, "json text needs " + std::to_string(s.size() + 1) + " bytes");
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return LDPC_OK;
}

int ldpc_code_as_json(const ldpc_code *code, char *buf, size_t capacity, size_t *needed)
try {
    if (!code || !needed) return fail(LDPC_ERR_BAD_ARG, "null argument");
    return give_text(linear_code_to_json(code->host), buf, capacity, needed);
}
LDPC_CATCH

int ldpc_code_from_json(const char *text, size_t len, ldpc_code **out)
try {
    if (!text || !out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    LinearCode c;
    std::string err;
    if (!linear_code_from_json(text, len, c, err)) return fail(LDPC_ERR_BAD_ARG, err);
    *out = new_code(std::move(c));
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_code_edges(const ldpc_code *code, size_t *bits, size_t *checks)
try {
    if (!code || ((!bits || !checks) && code->host.h.nnz())) return fail(LDPC_ERR_BAD_ARG, "null argument");
    const SparseBinMat &h = code->host.h;
    for (uint32_t r = 0; r < h.nrows(); r++)
        for (uint32_t e = h.ptr[r]; e < h.ptr[r + 1]; e++) {
            bits[e] = h.idx[e];
            checks[e] = r;
        }
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_code_syndrome_batch(const ldpc_code *code, size_t n_shots, const uint64_t *d_words, uint64_t *d_syndromes)
try {
    if (!code || (n_shots && (!d_words || !d_syndromes))) return fail(LDPC_ERR_BAD_ARG, "null argument");
    int st = code->ensure_device();
    if (st) return st;
    KernelTimer timer;
    HIP_TRY(launch_syndrome(code->dg, n_shots, d_words, d_syndromes, g_stream));
    timer.stop_and_store();
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_code_syndrome_of(const ldpc_code *code, size_t len, const size_t *positions, size_t n_positions,
                          size_t *out_positions, size_t *out_n)
try {
    if (!code || !out_n || (n_positions && !positions)) return fail(LDPC_ERR_BAD_ARG, "null argument");
    if (len != code->host.len())
        return fail(LDPC_ERR_LENGTH_MISMATCH, "message of length " + std::to_string(len) +
                                                  " is invalid for code with length " +
                                                  std::to_string(code->host.len()));   // mod.rs:397-403
    if (!valid_positions(positions, n_positions, len)) return fail(LDPC_ERR_BAD_ARG, "positions are not sorted, unique and in range");
    size_t m = code->host.num_checks();
    std::vector<uint64_t> in = pack(positions, n_positions, len), out(std::max<size_t>(words_for(m), 1), 0);
    int st = code->ensure_device();   // (before the staging buffers: they must live on the handle's device)
    if (st) return st;
    std::lock_guard<std::mutex> io(code->io_mu);
    DeviceBuffer &din = code->io_in, &dout = code->io_out;
    HIP_TRY(din.reserve(in.size() * 8));
    HIP_TRY(dout.reserve(out.size() * 8));
    HIP_TRY(hipMemcpyAsync(din.p, in.data(), in.size() * 8, hipMemcpyHostToDevice, g_stream));
    st = ldpc_code_syndrome_batch(code, 1, din.as<uint64_t>(), dout.as<uint64_t>());
    if (st) return st;
    HIP_TRY(hipMemcpyAsync(out.data(), dout.p, out.size() * 8, hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    if (!m) out[0] = 0;
    size_t k = 0;
    if (out_positions) unpack(out, m, out_positions, &k);
    else
        for (size_t i = 0; i < m; i++) k += (out[i >> 6] >> (i & 63)) & 1;
    *out_n = k;
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_code_has_codeword(const ldpc_code *code, size_t len, const size_t *positions, size_t n_positions, int *out)
try {
    if (!out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    size_t k = 0;
    int st = ldpc_code_syndrome_of(code, len, positions, n_positions, nullptr, &k);
    if (st) return st;
    *out = k == 0;   // mod.rs:435
    return LDPC_OK;
}
LDPC_CATCH

void ldpc_code_destroy(ldpc_code *code) { delete code; }

// ---- noise ----------------------------------------------------------------
int ldpc_bsc_sample_batch(double probability, size_t len, uint64_t seed, uint64_t first_shot, size_t n_shots,
                          uint64_t *d_words)
try {
    if (!valid_probability(probability)) return fail(LDPC_ERR_BAD_ARG, "probability is not between 0 and 1");
    if (!samplable_probability(probability)) return fail(LDPC_ERR_BAD_ARG, SAMPLED_PROBABILITY_MSG);
    if (len >= 0xFFFFFFFFull) return fail(LDPC_ERR_CAPACITY, "length exceeds 32 bits");
    if (n_shots && len && !d_words) return fail(LDPC_ERR_BAD_ARG, "null argument");
    KernelTimer timer;
    HIP_TRY(launch_sample_bsc(make_gap_table(probability), (uint32_t)len, seed, first_shot, n_shots, d_words, g_stream));
    timer.stop_and_store();
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_depolarizing_sample_batch(double probability, size_t len, uint64_t seed, uint64_t first_shot, size_t n_shots,
                                   uint64_t *d_x_words, uint64_t *d_z_words)
try {
    if (!valid_probability(probability)) return fail(LDPC_ERR_BAD_ARG, "probability is not between 0 and 1");
    if (!samplable_probability(probability)) return fail(LDPC_ERR_BAD_ARG, SAMPLED_PROBABILITY_MSG);
    if (len >= 0xFFFFFFFFull) return fail(LDPC_ERR_CAPACITY, "length exceeds 32 bits");
    if (n_shots && len && (!d_x_words || !d_z_words)) return fail(LDPC_ERR_BAD_ARG, "null argument");
    KernelTimer timer;
    HIP_TRY(launch_sample_depol(make_gap_table(probability), (uint32_t)len, seed, first_shot, n_shots, d_x_words,
                                d_z_words, g_stream));
    timer.stop_and_store();
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_bsc_sample_error_of_length(double probability, size_t len, uint64_t seed, uint64_t shot, size_t *out_positions,
                                    size_t *out_n)
try {
    if (!out_n || (len && !out_positions)) return fail(LDPC_ERR_BAD_ARG, "null argument");
    std::vector<uint64_t> w(std::max<size_t>(words_for(len), 1), 0);
    DeviceBuffer d;
    HIP_TRY(d.reserve(w.size() * 8));
    int st = ldpc_bsc_sample_batch(probability, len, seed, shot, 1, d.as<uint64_t>());
    if (st) return st;
    if (len) HIP_TRY(hipMemcpyAsync(w.data(), d.p, words_for(len) * 8, hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    unpack(w, len, out_positions, out_n);
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_erasure_sample_error_of_length(double probability, size_t len, uint64_t seed, uint64_t shot,
                                        size_t *out_positions, size_t *out_n)
try {
    // noise/erasure.rs:33-42 draws the same n i.i.d. Bernoulli(p) bits as the binary symmetric channel
    return ldpc_bsc_sample_error_of_length(probability, len, seed, shot, out_positions, out_n);
}
LDPC_CATCH

int ldpc_depolarizing_sample_error_of_length(double probability, size_t len, uint64_t seed, uint64_t shot,
                                             size_t *out_positions, uint8_t *out_paulis, size_t *out_n)
try {
    if (!out_n || (len && (!out_positions || !out_paulis))) return fail(LDPC_ERR_BAD_ARG, "null argument");
    size_t W = std::max<size_t>(words_for(len), 1);
    std::vector<uint64_t> x(W, 0), z(W, 0);
    DeviceBuffer dx, dz;
    HIP_TRY(dx.reserve(W * 8));
    HIP_TRY(dz.reserve(W * 8));
    int st = ldpc_depolarizing_sample_batch(probability, len, seed, shot, 1, dx.as<uint64_t>(), dz.as<uint64_t>());
    if (st) return st;
    if (len) {
        HIP_TRY(hipMemcpyAsync(x.data(), dx.p, words_for(len) * 8, hipMemcpyDeviceToHost, g_stream));
        HIP_TRY(hipMemcpyAsync(z.data(), dz.p, words_for(len) * 8, hipMemcpyDeviceToHost, g_stream));
    }
    HIP_TRY(hipStreamSynchronize(g_stream));
    size_t k = 0;
    for (size_t i = 0; i < len; i++) {
        int bx = (x[i >> 6] >> (i & 63)) & 1, bz = (z[i >> 6] >> (i & 63)) & 1;
        if (bx || bz) {
            out_positions[k] = i;
            out_paulis[k] = (uint8_t)(bx && bz ? 2 : (bx ? 1 : 3));
            k++;
        }
    }
    *out_n = k;
    return LDPC_OK;
}
LDPC_CATCH

// ---- FlipDecoder ----------------------------------------------------------
int ldpc_flip_create(const ldpc_code *code, ldpc_flip_decoder **out)
try {
    if (!code || !out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    ldpc_flip_decoder *d = new ldpc_flip_decoder();
    copy_code(d->code, code->host);
    *out = d;
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_flip_decode_batch(ldpc_flip_decoder *dec, size_t n_shots, const uint64_t *d_words_in, uint64_t *d_words_out,
                           uint64_t *d_syndromes_out, uint32_t *d_flips_out)
try {
    if (!dec || (n_shots && !d_words_in)) return fail(LDPC_ERR_BAD_ARG, "null argument");
    FlipArgs a{};
    a.source = SRC_WORDS;
    a.n_shots = n_shots;
    a.words_in = d_words_in;
    a.words_out = d_words_out;
    a.synd_out = d_syndromes_out;
    a.flips_out = d_flips_out;
    a.gap = make_gap_table(0.0);
    return run_flip(dec, a, nullptr);
}
LDPC_CATCH

int ldpc_flip_decode(ldpc_flip_decoder *dec, size_t len, const size_t *positions, size_t n_positions,
                     size_t *out_positions, size_t *out_n)
try {
    if (!dec || !out_n || (n_positions && !positions) || (len && !out_positions)) return fail(LDPC_ERR_BAD_ARG, "null argument");
    if (len != dec->code.host.len())
        return fail(LDPC_ERR_LENGTH_MISMATCH, "message of length " + std::to_string(len) +
                                                  " is invalid for code with length " +
                                                  std::to_string(dec->code.host.len()));
    if (!valid_positions(positions, n_positions, len)) return fail(LDPC_ERR_BAD_ARG, "positions are not sorted, unique and in range");
    std::vector<uint64_t> in = pack(positions, n_positions, len), out(in.size(), 0);
    int st = dec->code.ensure_device();
    if (st) return st;
    std::lock_guard<std::mutex> io(dec->io_mu);
    DeviceBuffer &din = dec->io_in, &dout = dec->io_out;
    HIP_TRY(din.reserve(in.size() * 8));
    HIP_TRY(dout.reserve(in.size() * 8));
    HIP_TRY(hipMemcpyAsync(din.p, in.data(), in.size() * 8, hipMemcpyHostToDevice, g_stream));
    st = ldpc_flip_decode_batch(dec, 1, din.as<uint64_t>(), dout.as<uint64_t>(), nullptr, nullptr);
    if (st) return st;
    HIP_TRY(hipMemcpyAsync(out.data(), dout.p, in.size() * 8, hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    if (!len) out[0] = 0;
    unpack(out, len, out_positions, out_n);
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_flip_mc_run(ldpc_flip_decoder *dec, double probability, uint64_t seed, uint64_t first_shot, size_t n_shots,
                     uint64_t counters[LDPC_COUNTERS], uint64_t *d_errors_out, uint64_t *d_syndromes_out,
                     uint64_t *d_words_out)
try {
    if (!dec || !counters) return fail(LDPC_ERR_BAD_ARG, "null argument");
    if (!valid_probability(probability)) return fail(LDPC_ERR_BAD_ARG, "probability is not between 0 and 1");
    if (!samplable_probability(probability)) return fail(LDPC_ERR_BAD_ARG, SAMPLED_PROBABILITY_MSG);
    FlipArgs a{};
    a.source = SRC_BSC;
    a.seed = seed;
    a.first_shot = first_shot;
    a.n_shots = n_shots;
    a.errors_out = d_errors_out;
    a.synd_out = d_syndromes_out;
    a.words_out = d_words_out;
    a.gap = make_gap_table(probability);
    a.expected_weight = (double)dec->code.host.len() * probability;
    a.list_cap = flip2_list_cap(a.expected_weight);   // error list sized for the channel
    return run_flip(dec, a, counters);
}
LDPC_CATCH

// The Monte-Carlo call for callers that cannot allocate (HIP graph capture, serving loops): the same kernels as
// ldpc_flip_mc_run with every piece of device scratch -- counters, parked input words, flip logs -- taken from
// the library's static device array (kernels.hip `scratch`, 64 MiB per GPU), no hipMalloc / hipFree, no event, no
// synchronisation.  The counters land in a device buffer of the caller when the stream reaches that point.
int ldpc_flip_mc_run_noalloc(ldpc_flip_decoder *dec, double probability, uint64_t seed, uint64_t first_shot, size_t n_shots,
                             uint64_t *d_counters_out, uint64_t *d_errors_out, uint64_t *d_syndromes_out, uint64_t *d_words_out)
try {
    if (!dec || !d_counters_out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    if (!valid_probability(probability)) return fail(LDPC_ERR_BAD_ARG, "probability is not between 0 and 1");
    if (!samplable_probability(probability)) return fail(LDPC_ERR_BAD_ARG, SAMPLED_PROBABILITY_MSG);
    std::lock_guard<std::mutex> lock(dec->mu);
    {
        std::lock_guard<std::mutex> code_lock(dec->code.mu);
        int current = 0;
        HIP_TRY(hipGetDevice(&current));
        if (!dec->code.uploaded || dec->path == 0)
            return fail(LDPC_ERR_BAD_ARG, "ldpc_flip_mc_run_noalloc: the handle is not set up on a device yet (graph and tables are "
                                          "uploaded by its first ordinary call: make one ldpc_flip_mc_run / ldpc_flip_decode_batch first)");
        if (current != dec->code.device)
            return fail(LDPC_ERR_BAD_ARG, "this handle's buffers live on device " + std::to_string(dec->code.device) +
                                              ", the calling thread's current device is " + std::to_string(current));
    }
    const DeviceGraph &g = dec->code.dg;
    unsigned char *base = nullptr;
    size_t bytes = 0;
    HIP_TRY(noalloc_scratch(&base, &bytes));
    // layout: 256 bytes of counters, then one piece per shot in flight (parked word + flip log), 8 pieces per block
    const size_t head = 256;
    FlipArgs a{};
    a.source = SRC_BSC;
    a.seed = seed;
    a.first_shot = first_shot;
    a.n_shots = n_shots;
    a.errors_out = d_errors_out;
    a.synd_out = d_syndromes_out;
    a.words_out = d_words_out;
    a.gap = make_gap_table(probability);
    a.expected_weight = (double)dec->code.host.len() * probability;
    a.max_flips = 0xFFFFFFFFu;
    a.counters = (unsigned long long *)base;
    a.flip_log_cap = ((g.m + 15u) & ~15u) + 16u;
    // (nothing is queued before every capacity check has passed: a refused call leaves the stream as it was)
    if (dec->path == 1) {
        const size_t per_wave = (words_for(g.m) + 2 * words_for(g.n)) * 8;
        if (33 * 8 + per_wave > 160 * 1024)
            return fail(LDPC_ERR_CAPACITY, "ldpc_flip_mc_run_noalloc: per-shot state of " + std::to_string(per_wave) +
                                               " bytes does not fit the LDS of a CU and the static array is not sized for it");
        HIP_TRY(hipMemsetAsync(base, 0, head, g_stream));   // the array keeps the last call's contents: the counters start at zero
        HIP_TRY(launch_flip(g, a, g_stream));
    } else {
        const size_t word_bytes = std::max<size_t>(words_for(g.n), 1) * 8;
        const size_t log_bytes = (size_t)a.flip_log_cap * (dec->path == 3 ? 2 : 4);
        const size_t per_block = 8 * ((word_bytes + 255) / 256 * 256 + (log_bytes + 255) / 256 * 256);
        const size_t blocks = (bytes - head) / per_block;
        if (blocks == 0)
            return fail(LDPC_ERR_CAPACITY, "ldpc_flip_mc_run_noalloc: one block's scratch of " + std::to_string(per_block) +
                                               " bytes exceeds the static array of " + std::to_string(bytes) + " bytes (code too large)");
        const uint32_t block_cap = (uint32_t)std::min<size_t>(blocks, 1u << 20);
        // parked words [block_cap * 8][word], then logs [block_cap * 8][log]: both strides as the kernels index them
        a.o_scratch = (uint64_t *)(base + head);
        const size_t log_off = (head + (size_t)block_cap * 8 * word_bytes + 255) & ~(size_t)255;
        a.flip_log = (uint32_t *)(base + log_off);
        if (log_off + (size_t)block_cap * 8 * log_bytes > bytes)
            return fail(LDPC_ERR_CAPACITY, "ldpc_flip_mc_run_noalloc: scratch layout exceeds the static array");
        HIP_TRY(hipMemsetAsync(base, 0, head, g_stream));   // the array keeps the last call's contents: the counters start at zero
        if (dec->path == 3) {
            a.list_cap = flip3_list_cap(a.expected_weight);
            HIP_TRY(launch_flip3(g, dec->nbt3, a, false, g_stream, block_cap));
        } else {
            a.list_cap = flip2_list_cap(a.expected_weight);
            HIP_TRY(launch_flip2(g, dec->nbt, a, g_stream, block_cap));
        }
    }
    HIP_TRY(hipMemcpyAsync(d_counters_out, base, sizeof(uint64_t) * LDPC_COUNTERS, hipMemcpyDeviceToDevice, g_stream));
    return LDPC_OK;
}
LDPC_CATCH

void ldpc_flip_destroy(ldpc_flip_decoder *dec) { delete dec; }

// ---- BpDecoder ------------------------------------------------------------
int ldpc_bp_create(const ldpc_code *code, double probability, size_t max_iterations, ldpc_bp_decoder **out)
try {
    if (!code || !out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    std::unique_ptr<ldpc_bp_decoder> d(new ldpc_bp_decoder());
    int st = init_bp(d.get(), code->host, probability, max_iterations);
    if (st) return st;
    *out = d.release();
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_bp_correction_for_batch(ldpc_bp_decoder *dec, size_t n_shots, const uint64_t *d_syndromes,
                                 uint64_t *d_corrections_out, uint32_t *d_iterations_out, double *d_posteriors_out)
try {
    if (!dec || (n_shots && !d_syndromes)) return fail(LDPC_ERR_BAD_ARG, "null argument");
    BpRun r;
    r.source = SRC_SYNDROMES;
    r.test = TEST_NONE;
    r.n_shots = n_shots;
    r.synd_in = d_syndromes;
    r.corr_out = d_corrections_out;
    r.iters_out = d_iterations_out;
    r.post_out = d_posteriors_out;
    r.gap = make_gap_table(0.0);
    return run_bp(dec, r, nullptr);
}
LDPC_CATCH

int ldpc_bp_correction_for_batch_messages(ldpc_bp_decoder *dec, size_t n_shots, const uint64_t *d_syndromes,
                                          uint64_t *d_corrections_out, uint32_t *d_iterations_out,
                                          double *d_posteriors_out, double *d_check_messages_out)
try {
    if (!dec || (n_shots && !d_syndromes)) return fail(LDPC_ERR_BAD_ARG, "null argument");
    BpRun r;
    r.source = SRC_SYNDROMES;
    r.test = TEST_NONE;
    r.n_shots = n_shots;
    r.synd_in = d_syndromes;
    r.corr_out = d_corrections_out;
    r.iters_out = d_iterations_out;
    r.post_out = d_posteriors_out;
    r.chk_out = d_check_messages_out;
    r.gap = make_gap_table(0.0);
    return run_bp(dec, r, nullptr);
}
LDPC_CATCH

int ldpc_bp_decode_batch(ldpc_bp_decoder *dec, size_t n_shots, const uint64_t *d_words_in, uint64_t *d_words_out,
                         uint32_t *d_iterations_out)
try {
    if (!dec || (n_shots && !d_words_in)) return fail(LDPC_ERR_BAD_ARG, "null argument");
    BpRun r;
    r.source = SRC_WORDS;
    r.test = TEST_NONE;
    r.xor_input = 1;
    r.n_shots = n_shots;
    r.words_in = d_words_in;
    r.corr_out = d_words_out;
    r.iters_out = d_iterations_out;
    r.gap = make_gap_table(0.0);
    return run_bp(dec, r, nullptr);
}
LDPC_CATCH

static int bp_single(ldpc_bp_decoder *dec, bool is_syndrome, size_t len, const size_t *positions, size_t n_positions,
                     size_t *out_positions, size_t *out_n)
{
    if (!dec || !out_n || (n_positions && !positions)) return fail(LDPC_ERR_BAD_ARG, "null argument");
    size_t n = dec->code.host.len(), m = dec->code.host.num_checks();
    size_t expect = is_syndrome ? m : n;
    if (len != expect)
        return fail(LDPC_ERR_LENGTH_MISMATCH, std::string(is_syndrome ? "syndrome" : "message") + " of length " +
                                                  std::to_string(len) + " is invalid for a decoder expecting " +
                                                  std::to_string(expect));
    if (!valid_positions(positions, n_positions, len)) return fail(LDPC_ERR_BAD_ARG, "positions are not sorted, unique and in range");
    if (n && !out_positions) return fail(LDPC_ERR_BAD_ARG, "null argument");
    std::vector<uint64_t> in = pack(positions, n_positions, len), out(std::max<size_t>(words_for(n), 1), 0);
    int st = dec->code.ensure_device();
    if (st) return st;
    std::lock_guard<std::mutex> io(dec->io_mu);
    DeviceBuffer &din = dec->io_in, &dout = dec->io_out;
    HIP_TRY(din.reserve(in.size() * 8));
    HIP_TRY(dout.reserve(out.size() * 8));
    HIP_TRY(hipMemcpyAsync(din.p, in.data(), in.size() * 8, hipMemcpyHostToDevice, g_stream));
    st = is_syndrome
                 ? ldpc_bp_correction_for_batch(dec, 1, din.as<uint64_t>(), dout.as<uint64_t>(), nullptr, nullptr)
                 : ldpc_bp_decode_batch(dec, 1, din.as<uint64_t>(), dout.as<uint64_t>(), nullptr);
    if (st) return st;
    HIP_TRY(hipMemcpyAsync(out.data(), dout.p, out.size() * 8, hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    if (!n) out[0] = 0;
    unpack(out, n, out_positions, out_n);
    return LDPC_OK;
}

int ldpc_bp_correction_for(ldpc_bp_decoder *dec, size_t syndrome_len, const size_t *syndrome_positions,
                           size_t n_positions, size_t *out_positions, size_t *out_n)
try {
    return bp_single(dec, true, syndrome_len, syndrome_positions, n_positions, out_positions, out_n);
}
LDPC_CATCH

int ldpc_bp_decode(ldpc_bp_decoder *dec, size_t len, const size_t *positions, size_t n_positions, size_t *out_positions,
                   size_t *out_n)
try {
    return bp_single(dec, false, len, positions, n_positions, out_positions, out_n);
}
LDPC_CATCH

int ldpc_bp_mc_run(ldpc_bp_decoder *dec, double noise_probability, uint64_t seed, uint64_t first_shot, size_t n_shots,
                   uint64_t counters[LDPC_COUNTERS], uint8_t *d_failed_out)
try {
    if (!dec || !counters) return fail(LDPC_ERR_BAD_ARG, "null argument");
    if (!valid_probability(noise_probability)) return fail(LDPC_ERR_BAD_ARG, "probability is not between 0 and 1");
    if (!samplable_probability(noise_probability)) return fail(LDPC_ERR_BAD_ARG, SAMPLED_PROBABILITY_MSG);
    BpRun r;
    r.source = SRC_BSC;
    r.test = TEST_RESIDUAL;
    r.seed = seed;
    r.first_shot = first_shot;
    r.n_shots = n_shots;
    r.failed_out = d_failed_out;
    r.fail_bit = 1;
    r.gap = make_gap_table(noise_probability);
    if (d_failed_out && n_shots) HIP_TRY(hipMemsetAsync(d_failed_out, 0, n_shots, g_stream));
    return run_bp(dec, r, counters);
}
LDPC_CATCH

void ldpc_bp_destroy(ldpc_bp_decoder *dec) { delete dec; }

// ---- CssCode --------------------------------------------------------------
static ldpc_css_code *new_css(const CssCode &c)
{
    ldpc_css_code *h = new ldpc_css_code();
    h->m[0].host = c.hx;
    h->m[1].host = c.hz;
    h->m[2].host = c.lx;
    h->m[3].host = c.lz;
    return h;
}

int ldpc_css_create(const ldpc_code *x_code, const ldpc_code *z_code, ldpc_css_code **out)
try {
    if (!x_code || !z_code || !out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    CssCode c;
    int e = css_try_new(x_code->host, z_code->host, c);
    if (e == 1)
        return fail(LDPC_ERR_CSS, "different x and z lengths: " + std::to_string(x_code->host.len()) + " & " +
                                      std::to_string(z_code->host.len()));   // css/mod.rs:375-376
    if (e == 2) return fail(LDPC_ERR_CSS, "codes are not orthogonal");       // css/mod.rs:378
    *out = new_css(c);
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_css_create_with_logicals(const ldpc_code *x_stabs, const ldpc_code *z_stabs, const ldpc_code *x_logicals,
                                  const ldpc_code *z_logicals, ldpc_css_code **out)
try {
    if (!x_stabs || !z_stabs || !x_logicals || !z_logicals || !out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    size_t n = x_stabs->host.len();
    if (z_stabs->host.len() != n || x_logicals->host.len() != n || z_logicals->host.len() != n)
        return fail(LDPC_ERR_CSS, "different x and z lengths");
    CssCode c;
    c.hx = x_stabs->host;
    c.hz = z_stabs->host;
    c.lx = x_logicals->host;
    c.lz = z_logicals->host;
    *out = new_css(c);
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_css_shor(ldpc_css_code **out)
try {
    if (!out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    std::string err;
    SparseBinMat hx, hz, lx, lz;
    SparseBinMat::from_rows(9, {{0, 1, 2, 3, 4, 5}, {3, 4, 5, 6, 7, 8}}, hx, err);
    SparseBinMat::from_rows(9, {{0, 1}, {1, 2}, {3, 4}, {4, 5}, {6, 7}, {7, 8}}, hz, err);
    SparseBinMat::from_rows(9, {{0, 1, 2}}, lx, err);
    SparseBinMat::from_rows(9, {{0, 3, 6}}, lz, err);
    CssCode c;
    c.hx = make_code(hx);
    c.hz = make_code(hz);
    c.lx = make_code(lx);
    c.lz = make_code(lz);
    *out = new_css(c);
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_css_steane(ldpc_css_code **out)
try {
    if (!out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    CssCode c;
    LinearCode h = hamming_code();
    if (css_try_new(h, h, c)) return fail(LDPC_ERR_CSS, "codes are not orthogonal");
    *out = new_css(c);
    return LDPC_OK;
}
LDPC_CATCH

static int hgp(const LinearCode &a, const LinearCode &b, ldpc_css_code **out)
{
    uint64_t nq = (uint64_t)a.len() * b.len() + (uint64_t)a.num_checks() * b.num_checks();
    uint64_t nnz = (uint64_t)a.len() * b.h.nnz() + (uint64_t)a.h.nnz() * b.num_checks();
    if (nq > 200000 || nnz > 50000000ull)
        return fail(LDPC_ERR_CAPACITY, "hypergraph product too large for the host-side logical-operator algebra");
    CssCode c;
    LinearCode xc = make_code(hypergraph_product_x_checks(a, b)), zc = make_code(hypergraph_product_z_checks(a, b));
    int e = css_try_new(xc, zc, c);
    if (e) return fail(LDPC_ERR_CSS, e == 1 ? "different x and z lengths" : "codes are not orthogonal");
    *out = new_css(c);
    return LDPC_OK;
}

int ldpc_css_hypergraph_product(const ldpc_code *first, const ldpc_code *second, ldpc_css_code **out)
try {
    if (!first || !second || !out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    return hgp(first->host, second->host, out);
}
LDPC_CATCH

int ldpc_css_toric(size_t distance, ldpc_css_code **out)
try {
    if (!out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    if (distance < 2 || distance > 400) return fail(LDPC_ERR_BAD_ARG, "invalid toric code distance");
    std::vector<std::vector<uint32_t>> rows;                      // css/mod.rs:85-88
    for (uint32_t c = 0; c + 1 < distance; c++) rows.push_back({c, c + 1});
    rows.push_back({0, (uint32_t)distance - 1});
    SparseBinMat h;
    std::string err;
    if (!SparseBinMat::from_rows((uint32_t)distance, rows, h, err)) return fail(LDPC_ERR_BAD_ARG, err);
    LinearCode code = make_code(h);
    return hgp(code, code, out);
}
LDPC_CATCH

size_t ldpc_css_len(const ldpc_css_code *css) { return css ? css->m[0].host.len() : 0; }

const ldpc_code *ldpc_css_matrix(const ldpc_css_code *css, int which)
{
    if (!css || which < 0 || which > 3) return nullptr;
    return &css->m[which];
}

int ldpc_css_syndrome_of(const ldpc_css_code *css, size_t len, const size_t *x_part, size_t n_x, const size_t *z_part,
                         size_t n_z, size_t *out_sx, size_t *out_n_sx, size_t *out_sz, size_t *out_n_sz)
try {
    if (!css || !out_n_sx || !out_n_sz) return fail(LDPC_ERR_BAD_ARG, "null argument");
    // css/mod.rs:200-205: stabilizers.pair(operator.swap_xz()): Hx * z_part, Hz * x_part
    int st = ldpc_code_syndrome_of(&css->m[0], len, z_part, n_z, out_sx, out_n_sx);
    if (st) return st;
    return ldpc_code_syndrome_of(&css->m[1], len, x_part, n_x, out_sz, out_n_sz);
}
LDPC_CATCH

int ldpc_css_has_logical(const ldpc_css_code *css, size_t len, const size_t *x_part, size_t n_x, const size_t *z_part,
                         size_t n_z, int *out)
try {
    if (!css || !out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    size_t a = 0, b = 0;
    int st = ldpc_css_syndrome_of(css, len, x_part, n_x, z_part, n_z, nullptr, &a, nullptr, &b);
    if (st) return st;
    *out = (a == 0 && b == 0);   // css.rs:137-139 is_trivial
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_css_has_stabilizer(const ldpc_css_code *css, size_t len, const size_t *x_part, size_t n_x,
                            const size_t *z_part, size_t n_z, int *out)
try {
    if (!css || !out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    int logical = 0;
    int st = ldpc_css_has_logical(css, len, x_part, n_x, z_part, n_z, &logical);
    if (st) return st;
    // css/mod.rs:249-251: every logical generator commutes with the operator:
    // X-type rows of lx against the z part, Z-type rows of lz against the x part
    size_t a = 0, b = 0;
    st = ldpc_code_syndrome_of(&css->m[2], len, z_part, n_z, nullptr, &a);
    if (st) return st;
    st = ldpc_code_syndrome_of(&css->m[3], len, x_part, n_x, nullptr, &b);
    if (st) return st;
    *out = logical && a == 0 && b == 0;
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_css_erasure_num_bad_errors(const ldpc_css_code *css, size_t len, const size_t *erasure, size_t n_erased,
                                    size_t *bad_x, size_t *bad_z)
try {
    if (!css || !bad_x || !bad_z || (n_erased && !erasure)) return fail(LDPC_ERR_BAD_ARG, "null argument");
    if (len != css->m[0].host.len())
        return fail(LDPC_ERR_LENGTH_MISMATCH, "erasure of length " + std::to_string(len) + " is invalid for code with length " +
                                                  std::to_string(css->m[0].host.len()));
    if (!valid_positions(erasure, n_erased, len)) return fail(LDPC_ERR_BAD_ARG, "positions are not sorted, unique and in range");
    std::vector<uint32_t> e(erasure, erasure + n_erased);
    // css_erasure.rs:20-34: X errors are seen by the Z stabilizers and Z logicals, Z errors by the X ones
    *bad_x = erasure_num_bad_errors(css->m[1].host.h, css->m[3].host.h, e);
    *bad_z = erasure_num_bad_errors(css->m[0].host.h, css->m[2].host.h, e);
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_css_erasure_recovery_probability(const ldpc_css_code *css, size_t len, const size_t *erasure, size_t n_erased,
                                          double *out)
try {
    if (!out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    size_t bx = 0, bz = 0;
    int st = ldpc_css_erasure_num_bad_errors(css, len, erasure, n_erased, &bx, &bz);
    if (st) return st;
    *out = std::ldexp(1.0, -(int)(bx + bz));   // css_erasure.rs:60-64
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_css_as_json(const ldpc_css_code *css, char *buf, size_t capacity, size_t *needed)
try {
    if (!css || !needed) return fail(LDPC_ERR_BAD_ARG, "null argument");
    CssCode c;
    c.hx = css->m[0].host;
    c.hz = css->m[1].host;
    c.lx = css->m[2].host;
    c.lz = css->m[3].host;
    return give_text(css_code_to_json(c), buf, capacity, needed);
}
LDPC_CATCH

int ldpc_css_from_json(const char *text, size_t len, ldpc_css_code **out)
try {
    if (!text || !out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    CssCode c;
    std::string err;
    if (!css_code_from_json(text, len, c, err))
        return fail(err.find("orthogonal") != std::string::npos || err.find("lengths") != std::string::npos ? LDPC_ERR_CSS
                                                                                                           : LDPC_ERR_BAD_ARG,
                    err);
    *out = new_css(c);
    return LDPC_OK;
}
LDPC_CATCH

void ldpc_css_destroy(ldpc_css_code *css) { delete css; }

// ---- CssDecoder<BpDecoder> -------------------------------------------------
int ldpc_css_bp_create(const ldpc_css_code *css, double probability, size_t max_iterations, ldpc_css_bp_decoder **out)
try {
    if (!css || !out) return fail(LDPC_ERR_BAD_ARG, "null argument");
    std::unique_ptr<ldpc_css_bp_decoder> d(new ldpc_css_bp_decoder());
    int st = init_bp(&d->x, css->m[0].host, probability, max_iterations);   // css_decoding.rs:42
    if (st) return st;
    st = init_bp(&d->z, css->m[1].host, probability, max_iterations);       // css_decoding.rs:43
    if (st) return st;
    copy_code(d->lx, css->m[2].host);
    copy_code(d->lz, css->m[3].host);
    *out = d.release();
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_css_bp_correction_for(ldpc_css_bp_decoder *dec, const size_t *sx, size_t n_sx, const size_t *sz, size_t n_sz,
                               size_t *out_cx, size_t *out_n_cx, size_t *out_cz, size_t *out_n_cz)
try {
    if (!dec) return fail(LDPC_ERR_BAD_ARG, "null argument");
    // decoders/css.rs:10-15: pair(x decoder, syndrome.x), pair(z decoder, syndrome.z), then swap_xz.  The two
    // decodes are independent and each is a handful of launch-bound rounds: the second runs from a host thread of
    // its own on the decoder's second stream, so that the rounds interleave.
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(dec->mu);
    if (!dec->side_stream) HIP_TRY(hipStreamCreateWithFlags(&dec->side_stream, hipStreamNonBlocking));
    int st_z = LDPC_OK;
    std::string err_z;
    auto z_half = [&]() {
        if (hipSetDevice(dev) != hipSuccess) {
            st_z = LDPC_ERR_HIP;
            err_z = "hipSetDevice failed";
            return;
        }
        g_stream = dec->side_stream;
        st_z = ldpc_bp_correction_for(&dec->z, dec->z.code.host.num_checks(), sz, n_sz, out_cx, out_n_cx);
        if (st_z) err_z = g_error;
    };
    std::thread other;
    bool beside = true;
    try {
        other = std::thread(z_half);
    } catch (...) {   // no thread to be had: one after the other (nothing may unwind across the C ABI)
        beside = false;
    }
    const int st = ldpc_bp_correction_for(&dec->x, dec->x.code.host.num_checks(), sx, n_sx, out_cz, out_n_cz);
    if (beside) other.join();
    else return st ? st : ldpc_bp_correction_for(&dec->z, dec->z.code.host.num_checks(), sz, n_sz, out_cx, out_n_cx);
    if (st) return st;
    if (st_z) return fail(st_z, err_z);
    return LDPC_OK;
}
LDPC_CATCH

int ldpc_css_bp_mc_run(ldpc_css_bp_decoder *dec, double noise_probability, uint64_t seed, uint64_t first_shot,
                       size_t n_shots, uint64_t counters[LDPC_COUNTERS])
try {
    if (!dec || !counters) return fail(LDPC_ERR_BAD_ARG, "null argument");
    if (!valid_probability(noise_probability)) return fail(LDPC_ERR_BAD_ARG, "probability is not between 0 and 1");
    if (!samplable_probability(noise_probability)) return fail(LDPC_ERR_BAD_ARG, SAMPLED_PROBABILITY_MSG);
    std::lock_guard<std::mutex> lock(dec->mu);
    for (int i = 0; i < LDPC_COUNTERS; i++) counters[i] = 0;
    GapTable gap = make_gap_table(noise_probability);
    const size_t chunk_max = (size_t)1 << 28;
    double ms_total = 0.0;
    uint64_t ran_total = 0;   // iterations actually run, both halves (ldpc_last_bp_executed_iterations)
    // Small runs (the streamed kernels: at most 128 groups of work) decode the two halves side by side, see below:
    // 1000 shots of the 625-qubit code 13.7 -> 7.3 ms.  Large runs do not: two persistent kernels on two streams were
    // seen to run one after the other, each on half of the workgroup slots (8.4e6 shots: 1.24 -> 1.65 s).
    bool side_by_side = n_shots <= 128 * 64;
    if (switches().css_side_by_side >= 0) side_by_side = switches().css_side_by_side != 0;   // A/B and test aid
    for (size_t done = 0; done < n_shots;) {
        size_t chunk = std::min(chunk_max, n_shots - done);
        HIP_TRY(dec->flags.reserve(chunk));
        HIP_TRY(hipMemsetAsync(dec->flags.p, 0, chunk, g_stream));
        uint64_t cx[LDPC_COUNTERS], cz[LDPC_COUNTERS], cf[LDPC_COUNTERS];
        BpRun r;
        r.test = TEST_LOGICAL;
        r.seed = seed;
        r.first_shot = first_shot + done;
        r.n_shots = chunk;
        r.gap = gap;
        // x part of the error: syndrome.z = Hz * e_x, decoded by the z decoder (BP on Hz);
        // the residual's x part must commute with the Z-type logicals (rows of lz)
        BpRun rz = r;
        rz.source = SRC_DEPOL_X;
        rz.fail_bit = 1;
        rz.logicals = &dec->lz;
        rz.failed_out = dec->flags.as<uint8_t>();
        // z part: syndrome.x = Hx * e_z, decoded by the x decoder (BP on Hx), logicals lx
        BpRun rx = r;
        rx.source = SRC_DEPOL_Z;
        rx.fail_bit = 2;
        rx.logicals = &dec->lx;
        rx.failed_out = dec->flags.as<uint8_t>();
        const uint8_t *flags2 = nullptr;
        if (side_by_side) {
            // The halves are independent (decoders/css.rs:10-15): run them at once, the second from a host thread
            // of its own, each on a stream of its own.  The launch-bound rounds of two small runs then interleave
            // instead of following each other.  Each half ORs into its own flag bytes (a byte has no atomic OR).
            HIP_TRY(dec->flags2.reserve(chunk));
            HIP_TRY(hipMemsetAsync(dec->flags2.p, 0, chunk, g_stream));
            HIP_TRY(hipStreamSynchronize(g_stream));
            rx.failed_out = dec->flags2.as<uint8_t>();
            flags2 = dec->flags2.as<uint8_t>();
            rz.share = rx.share = 2;
            int dev = 0;
            HIP_TRY(hipGetDevice(&dev));
            int st_x = LDPC_OK;
            std::string err_x;
            double ms_x = 0.0;
            uint64_t ran_x = 0;
            auto x_half = [&]() {
                hipStream_t sx = nullptr;
                if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&sx, hipStreamNonBlocking) != hipSuccess) {
                    st_x = LDPC_ERR_HIP;
                    err_x = "could not create the second stream";
                    return;
                }
                g_stream = sx;
                st_x = run_bp(&dec->x, rx, cz);
                if (st_x) err_x = g_error;
                ms_x = g_last_ms;
                ran_x = g_bp_executed;   // (this half's thread: thread-local)
                (void)hipStreamSynchronize(sx);
                (void)hipStreamDestroy(sx);
            };
            std::thread other;
            bool beside = true;
            try {
                other = std::thread(x_half);
            } catch (...) {   // no thread to be had: this thread runs the half itself, below
                beside = false;
            }
            // (this half too runs on a stream of its own: the null stream would order itself against the other)
            hipStream_t caller = g_stream, sz = nullptr;
            int st = LDPC_OK;
            if (hipStreamCreateWithFlags(&sz, hipStreamNonBlocking) != hipSuccess) {
                sz = nullptr;
                st = fail(LDPC_ERR_HIP, "could not create a stream");
            } else {
                g_stream = sz;
                st = run_bp(&dec->z, rz, cx);
                ran_total += g_bp_executed;
                (void)hipStreamSynchronize(sz);
                (void)hipStreamDestroy(sz);
                g_stream = caller;
            }
            const double ms_z = g_last_ms;
            if (beside) other.join();
            else if (!st) {
                hipStream_t caller2 = g_stream;
                x_half();            // sets g_stream to its own stream
                g_stream = caller2;
                ms_x += ms_z;        // in turn: the times add
            }
            if (st) return st;
            if (st_x) return fail(st_x, err_x);
            ran_total += ran_x;
            ms_total += std::max(ms_z, ms_x);
        } else {
            int st = run_bp(&dec->z, rz, cx);
            if (st) return st;
            ms_total += g_last_ms;
            ran_total += g_bp_executed;
            st = run_bp(&dec->x, rx, cz);
            if (st) return st;
            ms_total += g_last_ms;
            ran_total += g_bp_executed;
        }
        int st = 0;
        st = zero_counters(dec->counters);
        if (st) return st;
        HIP_TRY(launch_count_flags(dec->flags.as<uint8_t>(), flags2, chunk, dec->counters.as<unsigned long long>(), g_stream));
        st = read_counters(dec->counters, cf);
        if (st) return st;
        counters[LDPC_CNT_SHOTS] += chunk;
        counters[LDPC_CNT_FAILURES] += cf[LDPC_CNT_FAILURES];
        counters[LDPC_CNT_XFAIL] += cf[LDPC_CNT_XFAIL];
        counters[LDPC_CNT_ZFAIL] += cf[LDPC_CNT_ZFAIL];
        counters[LDPC_CNT_WORK] += cx[LDPC_CNT_WORK] + cz[LDPC_CNT_WORK];
        counters[LDPC_CNT_CAPPED] += cx[LDPC_CNT_CAPPED] + cz[LDPC_CNT_CAPPED];
        done += chunk;
    }
    g_last_ms = ms_total;
    g_bp_executed = ran_total;
    return LDPC_OK;
}
LDPC_CATCH

void ldpc_css_bp_destroy(ldpc_css_bp_decoder *dec) { delete dec; }

// ---- one process, several GPUs ---------------------------------------------
extern "C++" {
namespace {
// Runs `shard(k, first, count, counters)` on one host thread per device (device and stream are per-thread state)
// and sums the counters.  The first failing shard's status and message are reported.
template <class Shard>
int run_multi(int n_devices, const int *devices, uint64_t first_shot, size_t n_shots, uint64_t *counters, Shard shard)
{
    if (n_devices < 1 || !counters) return fail(LDPC_ERR_BAD_ARG, "n_devices must be at least 1");
    std::vector<std::array<uint64_t, LDPC_COUNTERS>> part((size_t)n_devices);
    std::vector<int> status((size_t)n_devices, LDPC_OK);
    std::vector<std::string> message((size_t)n_devices);
    std::vector<std::thread> threads;
    auto shard_k = [&](int k) {
            part[(size_t)k].fill(0);
            const size_t lo = (size_t)((unsigned __int128)n_shots * (unsigned)k / (unsigned)n_devices),
                         hi = (size_t)((unsigned __int128)n_shots * (unsigned)(k + 1) / (unsigned)n_devices);
            int st = ldpc_set_device(devices ? devices[k] : k);
            if (!st) st = shard(first_shot + lo, hi - lo, part[(size_t)k].data());
            status[(size_t)k] = st;
            if (st) message[(size_t)k] = ldpc_last_error();
    };
    bool spawned = true;
    for (int k = 0; k < n_devices && spawned; k++) {
        try {
            threads.emplace_back(shard_k, k);
        } catch (...) {   // nothing may unwind across the C ABI
            spawned = false;
        }
    }
    for (auto &t : threads) t.join();
    if (!spawned) return fail(LDPC_ERR_HIP, "could not start a host thread per device");
    for (int i = 0; i < LDPC_COUNTERS; i++) counters[i] = 0;
    for (int k = 0; k < n_devices; k++) {
        if (status[(size_t)k]) return fail(status[(size_t)k], "device shard " + std::to_string(k) + ": " + message[(size_t)k]);
        for (int i = 0; i < LDPC_COUNTERS; i++) counters[i] += part[(size_t)k][(size_t)i];
    }
    return LDPC_OK;
}
}  // namespace
}  // extern "C++"

int ldpc_flip_mc_run_multi(const ldpc_code *code, int n_devices, const int *devices, double probability, uint64_t seed,
                           uint64_t first_shot, size_t n_shots, uint64_t counters[LDPC_COUNTERS])
try {
    if (!code) return fail(LDPC_ERR_BAD_ARG, "null argument");
    return run_multi(n_devices, devices, first_shot, n_shots, counters, [&](uint64_t first, size_t count, uint64_t *c) {
        ldpc_flip_decoder *dec = nullptr;
        int st = ldpc_flip_create(code, &dec);
        if (!st) st = ldpc_flip_mc_run(dec, probability, seed, first, count, c, nullptr, nullptr, nullptr);
        ldpc_flip_destroy(dec);
        return st;
    });
}
LDPC_CATCH

int ldpc_bp_mc_run_multi(const ldpc_code *code, int n_devices, const int *devices, double probability,
                         size_t max_iterations, uint64_t seed, uint64_t first_shot, size_t n_shots,
                         uint64_t counters[LDPC_COUNTERS])
try {
    if (!code) return fail(LDPC_ERR_BAD_ARG, "null argument");
    return run_multi(n_devices, devices, first_shot, n_shots, counters, [&](uint64_t first, size_t count, uint64_t *c) {
        ldpc_bp_decoder *dec = nullptr;
        int st = ldpc_bp_create(code, probability, max_iterations, &dec);
        if (!st) st = ldpc_bp_mc_run(dec, probability, seed, first, count, c, nullptr);
        ldpc_bp_destroy(dec);
        return st;
    });
}
LDPC_CATCH

int ldpc_css_bp_mc_run_multi(const ldpc_css_code *css, int n_devices, const int *devices, double probability,
                             size_t max_iterations, uint64_t seed, uint64_t first_shot, size_t n_shots,
                             uint64_t counters[LDPC_COUNTERS])
try {
    if (!css) return fail(LDPC_ERR_BAD_ARG, "null argument");
    return run_multi(n_devices, devices, first_shot, n_shots, counters, [&](uint64_t first, size_t count, uint64_t *c) {
        ldpc_css_bp_decoder *dec = nullptr;
        int st = ldpc_css_bp_create(css, probability, max_iterations, &dec);
        if (!st) st = ldpc_css_bp_mc_run(dec, probability, seed, first, count, c);
        ldpc_css_bp_destroy(dec);
        return st;
    });
}
LDPC_CATCH

// Not part of the drop-in ABI (absent from include/ldpc_hip.h): lets the tests measure the
// accuracy of the BP kernel's tanh(x/2) / log1p routines (csrc/bp_math.hpp) on device arrays.
int ldpc_internal_bp_math(int kind, size_t n, const double *d_in, double *d_out)
try {
    if (n && (!d_in || !d_out)) return fail(LDPC_ERR_BAD_ARG, "null argument");
    HIP_TRY(launch_bp_math_probe(kind, n, d_in, d_out, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return LDPC_OK;
}
LDPC_CATCH

}  // extern "C"
