"""ctypes loader for the CPU oracle (oracle/ldpc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the `cpu_baseline` leg of bench.py, never by the product package `ldpc_amd`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libldpc_oracle.so")

CNT_SHOTS, CNT_FAIL, CNT_DETECTED, CNT_UNDETECTED, CNT_WORK, CNT_CAPPED, CNT_XFAIL, CNT_ZFAIL = range(8)


def build(force=False):
    """Compile the oracle with gcc if the .so is missing or stale."""
    src = os.path.join(_HERE, "ldpc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libldpc_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, u32, u64, i64, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64, C.c_double
        L.orc_code_new.restype = vp
        L.orc_code_new.argtypes = [u32, u32, vp, vp]
        L.orc_code_free.argtypes = [vp]
        L.orc_code_len.restype = u32
        L.orc_code_len.argtypes = [vp]
        L.orc_code_num_checks.restype = u32
        L.orc_code_num_checks.argtypes = [vp]
        L.orc_code_num_edges.restype = i64
        L.orc_code_num_edges.argtypes = [vp]
        L.orc_code_edges.argtypes = [vp, vp, vp]
        L.orc_code_bit_adjacencies.argtypes = [vp, vp, vp]
        L.orc_syndrome_pos.restype = i64
        L.orc_syndrome_pos.argtypes = [vp, u32, vp, C.c_size_t, vp]
        L.orc_flip_decode_pos.restype = i64
        L.orc_flip_decode_pos.argtypes = [vp, u32, vp, C.c_size_t, vp, vp]
        L.orc_syndrome_words.argtypes = [vp, vp, vp]
        L.orc_flip_decode_words.argtypes = [vp, vp, vp, vp, vp]
        L.orc_bp_correction_for.argtypes = [vp, dbl, u32, vp, vp, vp, vp, vp]
        L.orc_bp_decode_words.argtypes = [vp, dbl, u32, vp, vp, vp]
        L.orc_bp_correction_for_batch.argtypes = [vp, dbl, u32, u64, vp, vp, vp, vp, vp, C.c_int]
        L.orc_philox4x32_10.argtypes = [vp, vp, vp]
        L.orc_gap_table.argtypes = [dbl, vp]
        L.orc_sample_bsc_words.argtypes = [u64, u64, dbl, u32, vp]
        L.orc_sample_depolarizing_words.argtypes = [u64, u64, dbl, u32, vp, vp]
        L.orc_css_syndrome_words.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_css_residual_fails.restype = C.c_int
        L.orc_css_residual_fails.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_mc_flip.argtypes = [vp, dbl, u64, u64, u64, C.c_int, C.c_int, vp]
        L.orc_mc_bp.argtypes = [vp, dbl, u32, u64, u64, u64, C.c_int, vp]
        L.orc_mc_css_bp.argtypes = [vp, vp, vp, vp, dbl, u32, u64, u64, u64, C.c_int, vp]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def nwords(nbits):
    return (nbits + 63) // 64


def pos_to_words(pos, nbits):
    w = np.zeros(max(nwords(nbits), 1), dtype=np.uint64)
    for i in pos:
        w[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    return w[: nwords(nbits)] if nwords(nbits) else w[:0]


def words_to_pos(words, nbits):
    bits = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:nbits]
    return [int(i) for i in np.nonzero(bits)[0]]


class Code:
    """Sparse parity-check matrix handle (rows of ascending column indices)."""

    def __init__(self, n, rows=None, ptr=None, idx=None):
        if rows is not None:
            ptr = np.zeros(len(rows) + 1, dtype=np.int64)
            for i, r in enumerate(rows):
                ptr[i + 1] = ptr[i] + len(r)
            idx = np.array([c for r in rows for c in r], dtype=np.uint32)
        self.ptr = np.ascontiguousarray(ptr, dtype=np.int64)
        self.idx = np.ascontiguousarray(idx, dtype=np.uint32)
        if self.idx.size == 0:
            self.idx = np.zeros(0, dtype=np.uint32)
        self.n = int(n)
        self.m = len(self.ptr) - 1
        idx_arg = self.idx if self.idx.size else np.zeros(1, dtype=np.uint32)
        self.h = lib().orc_code_new(self.n, self.m, _p(self.ptr), _p(idx_arg))
        if not self.h:
            raise ValueError("invalid sparse binary matrix")
        self.E = int(lib().orc_code_num_edges(self.h))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_code_free(self.h)
                self.h = None
        except Exception:
            pass

    def rows(self):
        return [[int(x) for x in self.idx[self.ptr[i]: self.ptr[i + 1]]] for i in range(self.m)]

    def edges(self):
        bits = np.zeros(max(self.E, 1), dtype=np.uint32)
        checks = np.zeros(max(self.E, 1), dtype=np.uint32)
        lib().orc_code_edges(self.h, _p(bits), _p(checks))
        return [(int(b), int(c)) for b, c in zip(bits[: self.E], checks[: self.E])]

    def bit_adjacencies(self):
        ptr = np.zeros(self.n + 1, dtype=np.int64)
        idx = np.zeros(max(self.E, 1), dtype=np.uint32)
        lib().orc_code_bit_adjacencies(self.h, _p(ptr), _p(idx))
        return [[int(x) for x in idx[ptr[i]: ptr[i + 1]]] for i in range(self.n)]

    # --- position-list API (SparseBinVec style) ---
    def syndrome_of(self, pos, length=None):
        length = self.n if length is None else length
        a = np.array(sorted(pos), dtype=np.uint32)
        out = np.zeros(self.m + 1, dtype=np.uint32)
        k = lib().orc_syndrome_pos(self.h, length, _p(a if a.size else np.zeros(1, np.uint32)), a.size, _p(out))
        if k < 0:
            raise ValueError("length mismatch")
        return [int(x) for x in out[:k]]

    def flip_decode(self, pos, length=None):
        length = self.n if length is None else length
        a = np.array(sorted(pos), dtype=np.uint32)
        out = np.zeros(self.n + 1, dtype=np.uint32)
        flips = C.c_uint64(0)
        k = lib().orc_flip_decode_pos(self.h, length, _p(a if a.size else np.zeros(1, np.uint32)), a.size,
                                      _p(out), C.byref(flips))
        if k < 0:
            raise ValueError("length mismatch")
        return [int(x) for x in out[:k]], int(flips.value)

    # --- packed-word API ---
    def syndrome_words(self, e):
        e = np.ascontiguousarray(e, dtype=np.uint64)
        s = np.zeros(max(nwords(self.m), 1), dtype=np.uint64)
        lib().orc_syndrome_words(self.h, _p(e), _p(s))
        return s[: nwords(self.m)]

    def flip_decode_words(self, e):
        e = np.ascontiguousarray(e, dtype=np.uint64)
        out = np.zeros(max(nwords(self.n), 1), dtype=np.uint64)
        syn = np.zeros(max(nwords(self.m), 1), dtype=np.uint64)
        flips = C.c_uint64(0)
        lib().orc_flip_decode_words(self.h, _p(e), _p(out), _p(syn), C.byref(flips))
        return out[: nwords(self.n)], syn[: nwords(self.m)], int(flips.value)

    def bp_correction_for(self, p, max_iter, syn_words, want_messages=False):
        syn = np.ascontiguousarray(syn_words, dtype=np.uint64)
        if syn.size == 0:
            syn = np.zeros(1, dtype=np.uint64)
        corr = np.zeros(max(nwords(self.n), 1), dtype=np.uint64)
        it = C.c_uint32(0)
        post = np.zeros(max(self.n, 1), dtype=np.float64)
        chk = np.zeros(max(self.E, 1), dtype=np.float64)
        lib().orc_bp_correction_for(self.h, float(p), int(max_iter), _p(syn), _p(corr), C.byref(it), _p(post),
                                    _p(chk) if want_messages else None)
        if want_messages:
            return corr[: nwords(self.n)], int(it.value), post[: self.n], chk[: self.E]
        return corr[: nwords(self.n)], int(it.value), post[: self.n]

    def bp_correction_for_batch(self, p, max_iter, syn_words, want_post=False, want_messages=False, nthreads=1):
        """orc_bp_correction_for on every row of syn_words [shots][nwords(m)], shots spread over threads.
        Returns corrections [shots][nwords(n)], iterations [shots], posteriors or None, check messages or None."""
        syn = np.ascontiguousarray(syn_words, dtype=np.uint64)
        shots = syn.shape[0]
        corr = np.zeros((shots, max(nwords(self.n), 1)), dtype=np.uint64)
        its = np.zeros(shots, dtype=np.uint32)
        post = np.zeros((shots, max(self.n, 1)), dtype=np.float64) if want_post else None
        chk = np.zeros((shots, max(self.E, 1)), dtype=np.float64) if want_messages else None
        lib().orc_bp_correction_for_batch(self.h, float(p), int(max_iter), shots, _p(syn), _p(corr), _p(its),
                                          _p(post) if want_post else None, _p(chk) if want_messages else None, int(nthreads))
        return corr, its, post, chk

    def bp_decode_words(self, p, max_iter, msg_words):
        msg = np.ascontiguousarray(msg_words, dtype=np.uint64)
        out = np.zeros(max(nwords(self.n), 1), dtype=np.uint64)
        it = C.c_uint32(0)
        lib().orc_bp_decode_words(self.h, float(p), int(max_iter), _p(msg), _p(out), C.byref(it))
        return out[: nwords(self.n)], int(it.value)

    # --- Monte-Carlo ---
    def mc_flip(self, p, seed, first, nshots, reference_style=False, nthreads=1):
        cnt = np.zeros(8, dtype=np.uint64)
        lib().orc_mc_flip(self.h, float(p), seed, first, nshots, int(reference_style), nthreads, _p(cnt))
        return cnt

    def mc_bp(self, p, max_iter, seed, first, nshots, nthreads=1):
        cnt = np.zeros(8, dtype=np.uint64)
        lib().orc_mc_bp(self.h, float(p), int(max_iter), seed, first, nshots, nthreads, _p(cnt))
        return cnt


def philox(ctr, key):
    c = np.array(ctr, dtype=np.uint32)
    k = np.array(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(c), _p(k), _p(o))
    return [int(x) for x in o]


def gap_table(p):
    q = np.zeros(65, dtype=np.uint32)
    lib().orc_gap_table(float(p), _p(q))
    return q


def sample_bsc_words(seed, shot, p, n):
    w = np.zeros(max(nwords(n), 1), dtype=np.uint64)
    lib().orc_sample_bsc_words(seed, shot, float(p), n, _p(w))
    return w[: nwords(n)]


def sample_depolarizing_words(seed, shot, p, n):
    x = np.zeros(max(nwords(n), 1), dtype=np.uint64)
    z = np.zeros(max(nwords(n), 1), dtype=np.uint64)
    lib().orc_sample_depolarizing_words(seed, shot, float(p), n, _p(x), _p(z))
    return x[: nwords(n)], z[: nwords(n)]


def css_syndrome_words(hx, hz, ex, ez):
    sx = np.zeros(max(nwords(hx.m), 1), dtype=np.uint64)
    sz = np.zeros(max(nwords(hz.m), 1), dtype=np.uint64)
    lib().orc_css_syndrome_words(hx.h, hz.h, _p(np.ascontiguousarray(ex, np.uint64)),
                                 _p(np.ascontiguousarray(ez, np.uint64)), _p(sx), _p(sz))
    return sx[: nwords(hx.m)], sz[: nwords(hz.m)]


def css_residual_fails(hx, hz, lx, lz, rx, rz):
    """bit0: x part fails, bit1: z part fails; 0 <=> residual is a stabilizer."""
    return int(lib().orc_css_residual_fails(hx.h, hz.h, lx.h, lz.h, _p(np.ascontiguousarray(rx, np.uint64)),
                                            _p(np.ascontiguousarray(rz, np.uint64))))


def mc_css_bp(hx, hz, lx, lz, p, max_iter, seed, first, nshots, nthreads=1):
    cnt = np.zeros(8, dtype=np.uint64)
    lib().orc_mc_css_bp(hx.h, hz.h, lx.h, lz.h, float(p), int(max_iter), seed, first, nshots, nthreads, _p(cnt))
    return cnt


def max_threads():
    return int(lib().orc_max_threads())


# ---------------------------------------------------------------------------
# numpy restatements of the host-side algebra of the reference, used to check
# the product's host setup (small codes only).
# ---------------------------------------------------------------------------

def dense(rows, n):
    a = np.zeros((len(rows), n), dtype=np.uint8)
    for i, r in enumerate(rows):
        a[i, r] = 1
    return a


def rows_of(a):
    return [[int(j) for j in np.nonzero(r)[0]] for r in a]


def hypergraph_product(rows1, n1, rows2, n2):
    """CssCode::hypergraph_product (css/mod.rs:112-146):
    Hx = [I_n1 (x) H2 | H1^T (x) I_m2],  Hz = [H1 (x) I_n2 | I_m1 (x) H2^T]."""
    h1, h2 = dense(rows1, n1), dense(rows2, n2)
    m1, m2 = h1.shape[0], h2.shape[0]
    hx = np.hstack([np.kron(np.eye(n1, dtype=np.uint8), h2), np.kron(h1.T, np.eye(m2, dtype=np.uint8))])
    hz = np.hstack([np.kron(h1, np.eye(n2, dtype=np.uint8)), np.kron(np.eye(m1, dtype=np.uint8), h2.T)])
    return rows_of(hx), rows_of(hz), n1 * n2 + m1 * m2


def gf2_rank(a):
    a = a.copy() % 2
    r = 0
    rows, cols = a.shape
    for c in range(cols):
        piv = None
        for i in range(r, rows):
            if a[i, c]:
                piv = i
                break
        if piv is None:
            continue
        a[[r, piv]] = a[[piv, r]]
        for i in range(rows):
            if i != r and a[i, c]:
                a[i] ^= a[r]
        r += 1
        if r == rows:
            break
    return r
