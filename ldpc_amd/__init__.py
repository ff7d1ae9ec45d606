"""ldpc_amd -- host-side mirror of the `ldpc` crate's Monte-Carlo decoding API on MI355X.

Thin ctypes layer over the C ABI of include/ldpc_hip.h (ldpc_amd/libldpc_hip.so,
hand-written HIP kernels for gfx950).  Names follow the reference crate
(LinearCode, FlipDecoder, BpDecoder, CssCode, CssDecoder, BinarySymmetricChannel,
DepolarizingNoise, Probability) so tests read like the reference's own.

There is no CPU fallback: if the library is missing, importing the decoders
fails loudly; if no GPU is present, every compute call raises LdpcError.
Host-side code construction (matrices, random regular graphs, hypergraph
products, logical operators) works without a GPU.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libldpc_hip.so")

COUNTER_NAMES = ("shots", "failures", "detected", "undetected", "work", "capped", "xfail", "zfail")
(CNT_SHOTS, CNT_FAILURES, CNT_DETECTED, CNT_UNDETECTED, CNT_WORK, CNT_CAPPED, CNT_XFAIL, CNT_ZFAIL) = range(8)
N_COUNTERS = 8

OK, ERR_BAD_ARG, ERR_LENGTH_MISMATCH, ERR_SAMPLING, ERR_CSS, ERR_HIP, ERR_CAPACITY = range(7)


class LdpcError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("ldpc_hip status %d: %s" % (status, message))
        self.status = status


class LengthMismatch(LdpcError, ValueError):
    """The reference panics: linear_code/mod.rs:397-403."""


class SamplingError(LdpcError, ValueError):
    """random.rs:78-107"""


class CssError(LdpcError, ValueError):
    """css/mod.rs:366-381"""


_lib = None


def lib():
    """Load libldpc_hip.so.  torch (if importable) is imported first so that the
    process holds a single HIP runtime: the library then binds to the
    libamdhip64 torch already mapped."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C ldpc_amd/csrc` (or __graft_entry__.build()); "
            "there is no CPU fallback" % LIB_PATH)
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, u64, dbl, i = C.c_void_p, C.c_size_t, C.c_uint64, C.c_double, C.c_int
    pp = C.POINTER(vp)
    sig = {
        "ldpc_last_error": (C.c_char_p, []),
        "ldpc_device_count": (i, [C.POINTER(i)]),
        "ldpc_set_device": (i, [i]),
        "ldpc_set_stream": (i, [vp]),
        "ldpc_last_kernel_ms": (dbl, []),
        "ldpc_debug_reload_switches": (None, []),
        "ldpc_last_kernel_name": (C.c_char_p, []),
        "ldpc_last_bp_executed_iterations": (C.c_uint64, []),
        "ldpc_code_create": (i, [sz, sz, vp, vp, pp]),
        "ldpc_code_hamming": (i, [pp]),
        "ldpc_code_repetition": (i, [sz, pp]),
        "ldpc_code_random_regular": (i, [sz, sz, sz, sz, u64, pp]),
        "ldpc_code_len": (sz, [vp]),
        "ldpc_code_num_checks": (sz, [vp]),
        "ldpc_code_num_edges": (sz, [vp]),
        "ldpc_code_parity_check_matrix": (i, [vp, vp, vp]),
        "ldpc_code_bit_adjacencies": (i, [vp, vp, vp]),
        "ldpc_code_edges": (i, [vp, vp, vp]),
        "ldpc_code_syndrome_of": (i, [vp, sz, vp, sz, vp, C.POINTER(sz)]),
        "ldpc_code_has_codeword": (i, [vp, sz, vp, sz, C.POINTER(i)]),
        "ldpc_code_syndrome_batch": (i, [vp, sz, vp, vp]),
        "ldpc_code_num_generators": (i, [vp, C.POINTER(sz)]),
        "ldpc_code_generator_matrix": (i, [vp, vp, vp, sz]),
        "ldpc_code_dimension": (i, [vp, C.POINTER(sz)]),
        "ldpc_code_minimal_distance": (i, [vp, C.POINTER(sz), C.POINTER(i)]),
        "ldpc_code_has_same_codespace": (i, [vp, vp, C.POINTER(i)]),
        "ldpc_code_from_generator_matrix": (i, [sz, sz, vp, vp, pp]),
        "ldpc_code_from_both_matrices": (i, [sz, sz, vp, vp, sz, vp, vp, pp]),
        "ldpc_code_as_json": (i, [vp, vp, sz, C.POINTER(sz)]),
        "ldpc_code_from_json": (i, [C.c_char_p, sz, pp]),
        "ldpc_css_as_json": (i, [vp, vp, sz, C.POINTER(sz)]),
        "ldpc_css_from_json": (i, [C.c_char_p, sz, pp]),
        "ldpc_code_destroy": (None, [vp]),
        "ldpc_erasure_sample_error_of_length": (i, [dbl, sz, u64, u64, vp, C.POINTER(sz)]),
        "ldpc_bsc_sample_error_of_length": (i, [dbl, sz, u64, u64, vp, C.POINTER(sz)]),
        "ldpc_depolarizing_sample_error_of_length": (i, [dbl, sz, u64, u64, vp, vp, C.POINTER(sz)]),
        "ldpc_bsc_sample_batch": (i, [dbl, sz, u64, u64, sz, vp]),
        "ldpc_depolarizing_sample_batch": (i, [dbl, sz, u64, u64, sz, vp, vp]),
        "ldpc_flip_create": (i, [vp, pp]),
        "ldpc_flip_decode": (i, [vp, sz, vp, sz, vp, C.POINTER(sz)]),
        "ldpc_flip_decode_batch": (i, [vp, sz, vp, vp, vp, vp]),
        "ldpc_flip_mc_run": (i, [vp, dbl, u64, u64, sz, vp, vp, vp, vp]),
        "ldpc_flip_mc_run_noalloc": (i, [vp, dbl, u64, u64, sz, vp, vp, vp, vp]),
        "ldpc_flip_destroy": (None, [vp]),
        "ldpc_bp_create": (i, [vp, dbl, sz, pp]),
        "ldpc_bp_correction_for": (i, [vp, sz, vp, sz, vp, C.POINTER(sz)]),
        "ldpc_bp_decode": (i, [vp, sz, vp, sz, vp, C.POINTER(sz)]),
        "ldpc_bp_correction_for_batch": (i, [vp, sz, vp, vp, vp, vp]),
        "ldpc_bp_correction_for_batch_messages": (i, [vp, sz, vp, vp, vp, vp, vp]),
        "ldpc_bp_decode_batch": (i, [vp, sz, vp, vp, vp]),
        "ldpc_bp_mc_run": (i, [vp, dbl, u64, u64, sz, vp, vp]),
        "ldpc_bp_destroy": (None, [vp]),
        "ldpc_css_create": (i, [vp, vp, pp]),
        "ldpc_css_create_with_logicals": (i, [vp, vp, vp, vp, pp]),
        "ldpc_css_shor": (i, [pp]),
        "ldpc_css_steane": (i, [pp]),
        "ldpc_css_toric": (i, [sz, pp]),
        "ldpc_css_hypergraph_product": (i, [vp, vp, pp]),
        "ldpc_css_len": (sz, [vp]),
        "ldpc_css_matrix": (vp, [vp, i]),
        "ldpc_css_syndrome_of": (i, [vp, sz, vp, sz, vp, sz, vp, C.POINTER(sz), vp, C.POINTER(sz)]),
        "ldpc_css_has_logical": (i, [vp, sz, vp, sz, vp, sz, C.POINTER(i)]),
        "ldpc_css_has_stabilizer": (i, [vp, sz, vp, sz, vp, sz, C.POINTER(i)]),
        "ldpc_css_erasure_num_bad_errors": (i, [vp, sz, vp, sz, C.POINTER(sz), C.POINTER(sz)]),
        "ldpc_css_erasure_recovery_probability": (i, [vp, sz, vp, sz, C.POINTER(dbl)]),
        "ldpc_css_destroy": (None, [vp]),
        "ldpc_css_bp_create": (i, [vp, dbl, sz, pp]),
        "ldpc_css_bp_correction_for": (i, [vp, vp, sz, vp, sz, vp, C.POINTER(sz), vp, C.POINTER(sz)]),
        "ldpc_css_bp_mc_run": (i, [vp, dbl, u64, u64, sz, vp]),
        "ldpc_css_bp_destroy": (None, [vp]),
        "ldpc_flip_mc_run_multi": (i, [vp, i, vp, dbl, u64, u64, sz, vp]),
        "ldpc_bp_mc_run_multi": (i, [vp, i, vp, dbl, sz, u64, u64, sz, vp]),
        "ldpc_css_bp_mc_run_multi": (i, [vp, i, vp, dbl, sz, u64, u64, sz, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._signatures = sig
    _lib = L
    return L


def _check(status):
    if status == OK:
        return
    msg = lib().ldpc_last_error().decode("utf-8", "replace")
    cls = {ERR_LENGTH_MISMATCH: LengthMismatch, ERR_SAMPLING: SamplingError, ERR_CSS: CssError}.get(status, LdpcError)
    raise cls(status, msg)


def _szarr(values):
    a = np.ascontiguousarray(values, dtype=np.uintp)
    return a, (a.ctypes.data_as(C.c_void_p) if a.size else None)


def _ptr(x):
    """Device pointer from an int, a torch tensor, or None."""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    return C.c_void_p(x.data_ptr())


def nwords(nbits):
    return (nbits + 63) // 64


def device_count():
    c = C.c_int(0)
    st = lib().ldpc_device_count(C.byref(c))
    return c.value if st == OK else 0


def set_device(device):
    _check(lib().ldpc_set_device(int(device)))


def last_kernel_ms():
    return float(lib().ldpc_last_kernel_ms())


def last_kernel_name():
    return lib().ldpc_last_kernel_name().decode("ascii", "replace")


def last_bp_executed_iterations():
    """BP iterations the last mc_run of this thread actually ran (<= its CNT_WORK: periodic runs are cut short)."""
    return int(lib().ldpc_last_bp_executed_iterations())


def reload_switches():
    """The library reads its LDPC_HIP_* diagnostic switches once; call this after changing one (tests, A/B scripts)."""
    lib().ldpc_debug_reload_switches()


class Probability:
    """noise/mod.rs:32-51"""

    def __init__(self, probability):
        if not (0.0 <= probability <= 1.0):
            raise ValueError("probability is not between 0 and 1")
        self._p = float(probability)

    @staticmethod
    def try_new(probability):
        return Probability(probability) if 0.0 <= probability <= 1.0 else None

    def value(self):
        return self._p

    def __float__(self):
        return self._p

    def __repr__(self):
        return str(self._p)


def _pval(p):
    return p.value() if isinstance(p, Probability) else float(p)


def _rows_from_csr(ptr, idx):
    return [[int(x) for x in idx[ptr[r]: ptr[r + 1]]] for r in range(len(ptr) - 1)]


def _json_text(fn, handle):
    need = C.c_size_t(0)
    fn(handle, None, 0, C.byref(need))          # sizing call: LDPC_ERR_CAPACITY with *needed set
    if need.value == 0:
        _check(fn(handle, None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    _check(fn(handle, C.cast(buf, C.c_void_p), need.value, C.byref(need)))
    return buf.value.decode()


class LinearCode:
    """codes/linear_code/mod.rs:52-56 (parity-check matrix + bit adjacencies; the
    generator matrix is not built)."""

    def __init__(self, handle, owned=True):
        self._h = handle
        self._owned = owned

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "_h", None):
            try:
                lib().ldpc_code_destroy(self._h)
            except Exception:
                pass
            self._h = None

    # constructors -----------------------------------------------------------
    @staticmethod
    def from_parity_check_matrix(num_bits, rows):
        ptr = np.zeros(len(rows) + 1, dtype=np.uintp)
        for r, row in enumerate(rows):
            ptr[r + 1] = ptr[r] + len(row)
        idx = np.array([c for row in rows for c in row], dtype=np.uintp)
        return LinearCode.from_csr(num_bits, ptr, idx)

    @staticmethod
    def from_csr(num_bits, row_ptr, col_idx):
        ptr, pptr = _szarr(row_ptr)
        idx, pidx = _szarr(col_idx)
        h = C.c_void_p()
        _check(lib().ldpc_code_create(int(num_bits), len(ptr) - 1, pptr, pidx, C.byref(h)))
        return LinearCode(h)

    @staticmethod
    def _csr(rows):
        ptr = np.zeros(len(rows) + 1, dtype=np.uintp)
        for r, row in enumerate(rows):
            ptr[r + 1] = ptr[r] + len(row)
        idx = np.array([c for row in rows for c in row], dtype=np.uintp)
        return ptr, idx

    @staticmethod
    def from_generator_matrix(num_bits, rows):
        """linear_code/mod.rs:116-124"""
        ptr, idx = LinearCode._csr(rows)
        h = C.c_void_p()
        _check(lib().ldpc_code_from_generator_matrix(int(num_bits), len(rows), ptr.ctypes.data_as(C.c_void_p),
                                                     idx.ctypes.data_as(C.c_void_p), C.byref(h)))
        return LinearCode(h)

    @staticmethod
    def from_both_matrices(num_bits, generator_rows, parity_check_rows):
        """linear_code/mod.rs:59-72 (raises where the reference panics)"""
        gp, gi = LinearCode._csr(generator_rows)
        hp, hi = LinearCode._csr(parity_check_rows)
        h = C.c_void_p()
        _check(lib().ldpc_code_from_both_matrices(int(num_bits), len(generator_rows), gp.ctypes.data_as(C.c_void_p),
                                                  gi.ctypes.data_as(C.c_void_p), len(parity_check_rows),
                                                  hp.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p), C.byref(h)))
        return LinearCode(h)

    @staticmethod
    def empty():
        """linear_code/mod.rs:172-175: length 0, no checks"""
        return LinearCode.from_parity_check_matrix(0, [])

    @staticmethod
    def hamming_code():
        h = C.c_void_p()
        _check(lib().ldpc_code_hamming(C.byref(h)))
        return LinearCode(h)

    @staticmethod
    def repetition_code(length):
        h = C.c_void_p()
        _check(lib().ldpc_code_repetition(int(length), C.byref(h)))
        return LinearCode(h)

    @staticmethod
    def random_regular_code(num_bits, num_checks, bit_degree, check_degree, seed):
        """RandomRegularCode builder + sample_with (random.rs:23-65); `seed` stands for the rng."""
        h = C.c_void_p()
        _check(lib().ldpc_code_random_regular(int(num_bits), int(num_checks), int(bit_degree), int(check_degree),
                                              int(seed), C.byref(h)))
        return LinearCode(h)

    # accessors --------------------------------------------------------------
    def len(self):
        return int(lib().ldpc_code_len(self._h))

    __len__ = len

    def num_checks(self):
        return int(lib().ldpc_code_num_checks(self._h))

    def num_edges(self):
        return int(lib().ldpc_code_num_edges(self._h))

    def csr(self):
        ptr = np.zeros(self.num_checks() + 1, dtype=np.uintp)
        idx = np.zeros(max(self.num_edges(), 1), dtype=np.uintp)
        _check(lib().ldpc_code_parity_check_matrix(self._h, ptr.ctypes.data_as(C.c_void_p),
                                                   idx.ctypes.data_as(C.c_void_p)))
        return ptr, idx[: self.num_edges()]

    def parity_check_matrix(self):
        return _rows_from_csr(*self.csr())

    def bit_adjacencies(self):
        ptr = np.zeros(self.len() + 1, dtype=np.uintp)
        idx = np.zeros(max(self.num_edges(), 1), dtype=np.uintp)
        _check(lib().ldpc_code_bit_adjacencies(self._h, ptr.ctypes.data_as(C.c_void_p),
                                               idx.ctypes.data_as(C.c_void_p)))
        return _rows_from_csr(ptr, idx)

    def edges(self):
        """(bit, check) pairs in the reference's Edges order (edges.rs:23-29)."""
        e = self.num_edges()
        bits = np.zeros(max(e, 1), dtype=np.uintp)
        checks = np.zeros(max(e, 1), dtype=np.uintp)
        _check(lib().ldpc_code_edges(self._h, bits.ctypes.data_as(C.c_void_p), checks.ctypes.data_as(C.c_void_p)))
        return [(int(b), int(c)) for b, c in zip(bits[:e], checks[:e])]

    # host algebra (small codes) ---------------------------------------------
    def num_generators(self):
        """linear_code/mod.rs:290-292"""
        k = C.c_size_t(0)
        _check(lib().ldpc_code_num_generators(self._h, C.byref(k)))
        return k.value

    def dimension(self):
        """linear_code/mod.rs:308-310"""
        k = C.c_size_t(0)
        _check(lib().ldpc_code_dimension(self._h, C.byref(k)))
        return k.value

    def generator_matrix(self):
        """Rows of a basis of the codespace (linear_code/mod.rs:225-227); the basis itself is this library's."""
        k, n = self.num_generators(), self.len()
        ptr = np.zeros(k + 1, dtype=np.uintp)
        idx = np.zeros(max(k * n, 1), dtype=np.uintp)
        _check(lib().ldpc_code_generator_matrix(self._h, ptr.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p),
                                                idx.size))
        return _rows_from_csr(ptr, idx)

    def minimal_distance(self):
        """linear_code/mod.rs:318-335: None if the code has no non-trivial codeword."""
        d, found = C.c_size_t(0), C.c_int(0)
        _check(lib().ldpc_code_minimal_distance(self._h, C.byref(d), C.byref(found)))
        return d.value if found.value else None

    def has_same_codespace(self, other):
        """linear_code/mod.rs:272-275"""
        r = C.c_int(0)
        _check(lib().ldpc_code_has_same_codespace(self._h, other._h, C.byref(r)))
        return bool(r.value)

    def as_json(self):
        """linear_code/mod.rs:468-471 (layout of the inner matrices: see include/ldpc_hip.h, parity unpinned)"""
        return _json_text(lib().ldpc_code_as_json, self._h)

    @staticmethod
    def from_json(text):
        """serde_json::from_str::<LinearCode> (derive linear_code/mod.rs:51)"""
        raw = text.encode() if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        _check(lib().ldpc_code_from_json(raw, len(raw), C.byref(h)))
        return LinearCode(h)

    # hot path ---------------------------------------------------------------
    def syndrome_of(self, positions, length=None):
        length = self.len() if length is None else length
        pos, ppos = _szarr(positions)
        out = np.zeros(max(self.num_checks(), 1), dtype=np.uintp)
        k = C.c_size_t(0)
        _check(lib().ldpc_code_syndrome_of(self._h, length, ppos, pos.size, out.ctypes.data_as(C.c_void_p),
                                           C.byref(k)))
        return [int(x) for x in out[: k.value]]

    def has_codeword(self, positions, length=None):
        length = self.len() if length is None else length
        pos, ppos = _szarr(positions)
        r = C.c_int(0)
        _check(lib().ldpc_code_has_codeword(self._h, length, ppos, pos.size, C.byref(r)))
        return bool(r.value)

    def syndrome_batch(self, n_shots, d_words, d_syndromes):
        _check(lib().ldpc_code_syndrome_batch(self._h, int(n_shots), _ptr(d_words), _ptr(d_syndromes)))

    def random_error(self, noise_model, seed, shot):
        """linear_code/mod.rs:460-466"""
        return noise_model.sample_error_of_length(self.len(), seed, shot)


class BinarySymmetricChannel:
    """noise/binary_symmetric_channel.rs:15-44"""

    def __init__(self, probability):
        self.probability = Probability(_pval(probability)).value()

    @staticmethod
    def with_probability(probability):
        return BinarySymmetricChannel(probability)

    def sample_error_of_length(self, length, seed, shot):
        out = np.zeros(max(length, 1), dtype=np.uintp)
        k = C.c_size_t(0)
        _check(lib().ldpc_bsc_sample_error_of_length(self.probability, int(length), int(seed), int(shot),
                                                     out.ctypes.data_as(C.c_void_p), C.byref(k)))
        return [int(x) for x in out[: k.value]]

    def sample_batch(self, length, seed, first_shot, n_shots, d_words):
        _check(lib().ldpc_bsc_sample_batch(self.probability, int(length), int(seed), int(first_shot), int(n_shots),
                                           _ptr(d_words)))

    def __str__(self):
        return "BSC(%s)" % self.probability


class ErasureChannel:
    """noise/erasure.rs:13-49; the sample is the list of erased positions."""

    def __init__(self, probability):
        self.probability = _pval(probability)

    @staticmethod
    def with_probability(probability):
        return ErasureChannel(probability)

    def sample_error_of_length(self, length, seed, shot):
        out = np.zeros(max(length, 1), dtype=np.uintp)
        k = C.c_size_t(0)
        _check(lib().ldpc_erasure_sample_error_of_length(self.probability, int(length), int(seed), int(shot),
                                                         out.ctypes.data_as(C.c_void_p), C.byref(k)))
        return [int(x) for x in out[: k.value]]

    def __str__(self):
        return "Erasure(%s)" % self.probability


class DepolarizingNoise:
    """noise/depolarizing.rs:12-49"""

    PAULIS = {1: "X", 2: "Y", 3: "Z"}

    def __init__(self, probability):
        self.probability = Probability(_pval(probability)).value()

    @staticmethod
    def with_probability(probability):
        return DepolarizingNoise(probability)

    def sample_error_of_length(self, length, seed, shot):
        """Returns (positions, paulis) with paulis in 'X', 'Y', 'Z'."""
        out = np.zeros(max(length, 1), dtype=np.uintp)
        pl = np.zeros(max(length, 1), dtype=np.uint8)
        k = C.c_size_t(0)
        _check(lib().ldpc_depolarizing_sample_error_of_length(self.probability, int(length), int(seed), int(shot),
                                                              out.ctypes.data_as(C.c_void_p),
                                                              pl.ctypes.data_as(C.c_void_p), C.byref(k)))
        return [int(x) for x in out[: k.value]], [self.PAULIS[int(p)] for p in pl[: k.value]]

    def sample_batch(self, length, seed, first_shot, n_shots, d_x_words, d_z_words):
        _check(lib().ldpc_depolarizing_sample_batch(self.probability, int(length), int(seed), int(first_shot),
                                                    int(n_shots), _ptr(d_x_words), _ptr(d_z_words)))

    def __str__(self):
        return "Depolarizing Noise (prob = %s)" % self.probability


class FlipDecoder:
    """decoders/flip.rs:7-47"""

    def __init__(self, code):
        self.code = code
        self._h = C.c_void_p()
        _check(lib().ldpc_flip_create(code._h, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().ldpc_flip_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def decode(self, positions, length=None):
        length = self.code.len() if length is None else length
        pos, ppos = _szarr(positions)
        out = np.zeros(max(self.code.len(), 1), dtype=np.uintp)
        k = C.c_size_t(0)
        _check(lib().ldpc_flip_decode(self._h, length, ppos, pos.size, out.ctypes.data_as(C.c_void_p), C.byref(k)))
        return [int(x) for x in out[: k.value]]

    def decode_batch(self, n_shots, d_words_in, d_words_out=None, d_syndromes_out=None, d_flips_out=None):
        _check(lib().ldpc_flip_decode_batch(self._h, int(n_shots), _ptr(d_words_in), _ptr(d_words_out),
                                            _ptr(d_syndromes_out), _ptr(d_flips_out)))

    def mc_run(self, probability, seed, first_shot, n_shots, d_errors_out=None, d_syndromes_out=None,
               d_words_out=None):
        cnt = np.zeros(N_COUNTERS, dtype=np.uint64)
        _check(lib().ldpc_flip_mc_run(self._h, _pval(probability), int(seed), int(first_shot), int(n_shots),
                                      cnt.ctypes.data_as(C.c_void_p), _ptr(d_errors_out), _ptr(d_syndromes_out),
                                      _ptr(d_words_out)))
        return cnt

    def mc_run_noalloc(self, probability, seed, first_shot, n_shots, d_counters_out, d_errors_out=None,
                       d_syndromes_out=None, d_words_out=None):
        """ldpc_flip_mc_run_noalloc: no device allocation, no synchronisation; the counters go to the device buffer
        d_counters_out ([8] u64) and are valid after the stream has been synchronised."""
        _check(lib().ldpc_flip_mc_run_noalloc(self._h, _pval(probability), int(seed), int(first_shot), int(n_shots),
                                              _ptr(d_counters_out), _ptr(d_errors_out), _ptr(d_syndromes_out),
                                              _ptr(d_words_out)))

    def __str__(self):
        return "Flip decoder"


class BpDecoder:
    """decoders/belief_propagation.rs:8-89"""

    def __init__(self, code, probability, num_iterations):
        self.code = code
        self.probability = Probability(_pval(probability)).value()
        self.num_iterations = int(num_iterations)
        self._h = C.c_void_p()
        _check(lib().ldpc_bp_create(code._h, self.probability, self.num_iterations, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().ldpc_bp_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def num_bits(self):
        return self.code.len()

    def num_checks(self):
        return self.code.num_checks()

    def correction_for(self, syndrome_positions, length=None):
        length = self.code.num_checks() if length is None else length
        pos, ppos = _szarr(syndrome_positions)
        out = np.zeros(max(self.code.len(), 1), dtype=np.uintp)
        k = C.c_size_t(0)
        _check(lib().ldpc_bp_correction_for(self._h, length, ppos, pos.size, out.ctypes.data_as(C.c_void_p),
                                            C.byref(k)))
        return [int(x) for x in out[: k.value]]

    def decode(self, positions, length=None):
        length = self.code.len() if length is None else length
        pos, ppos = _szarr(positions)
        out = np.zeros(max(self.code.len(), 1), dtype=np.uintp)
        k = C.c_size_t(0)
        _check(lib().ldpc_bp_decode(self._h, length, ppos, pos.size, out.ctypes.data_as(C.c_void_p), C.byref(k)))
        return [int(x) for x in out[: k.value]]

    def correction_for_batch(self, n_shots, d_syndromes, d_corrections_out=None, d_iterations_out=None,
                             d_posteriors_out=None, d_check_messages_out=None):
        if d_check_messages_out is not None:   # [n_shots][E] f64: check -> bit messages at stop (parity output)
            _check(lib().ldpc_bp_correction_for_batch_messages(self._h, int(n_shots), _ptr(d_syndromes), _ptr(d_corrections_out),
                                                               _ptr(d_iterations_out), _ptr(d_posteriors_out),
                                                               _ptr(d_check_messages_out)))
            return
        _check(lib().ldpc_bp_correction_for_batch(self._h, int(n_shots), _ptr(d_syndromes), _ptr(d_corrections_out),
                                                  _ptr(d_iterations_out), _ptr(d_posteriors_out)))

    def decode_batch(self, n_shots, d_words_in, d_words_out=None, d_iterations_out=None):
        _check(lib().ldpc_bp_decode_batch(self._h, int(n_shots), _ptr(d_words_in), _ptr(d_words_out),
                                          _ptr(d_iterations_out)))

    def mc_run(self, noise_probability, seed, first_shot, n_shots, d_failed_out=None):
        cnt = np.zeros(N_COUNTERS, dtype=np.uint64)
        _check(lib().ldpc_bp_mc_run(self._h, _pval(noise_probability), int(seed), int(first_shot), int(n_shots),
                                    cnt.ctypes.data_as(C.c_void_p), _ptr(d_failed_out)))
        return cnt


class CssCode:
    """codes/css/mod.rs:25-28"""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().ldpc_css_destroy(self._h)
            except Exception:
                pass
            self._h = None

    @staticmethod
    def new(x_code, z_code):
        h = C.c_void_p()
        _check(lib().ldpc_css_create(x_code._h, z_code._h, C.byref(h)))
        return CssCode(h)

    try_new = new

    @staticmethod
    def with_logicals(x_stabs, z_stabs, x_logicals, z_logicals):
        h = C.c_void_p()
        _check(lib().ldpc_css_create_with_logicals(x_stabs._h, z_stabs._h, x_logicals._h, z_logicals._h, C.byref(h)))
        return CssCode(h)

    @staticmethod
    def shor_code():
        h = C.c_void_p()
        _check(lib().ldpc_css_shor(C.byref(h)))
        return CssCode(h)

    @staticmethod
    def steane_code():
        h = C.c_void_p()
        _check(lib().ldpc_css_steane(C.byref(h)))
        return CssCode(h)

    @staticmethod
    def toric_code(distance):
        h = C.c_void_p()
        _check(lib().ldpc_css_toric(int(distance), C.byref(h)))
        return CssCode(h)

    @staticmethod
    def hypergraph_product(first_code, second_code):
        h = C.c_void_p()
        _check(lib().ldpc_css_hypergraph_product(first_code._h, second_code._h, C.byref(h)))
        return CssCode(h)

    def as_json(self):
        """serde form of CssCode (derive codes/css/mod.rs:24-28)"""
        return _json_text(lib().ldpc_css_as_json, self._h)

    @staticmethod
    def from_json(text):
        raw = text.encode() if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        _check(lib().ldpc_css_from_json(raw, len(raw), C.byref(h)))
        return CssCode(h)

    def len(self):
        return int(lib().ldpc_css_len(self._h))

    __len__ = len

    def _matrix(self, which):
        code = LinearCode(C.c_void_p(lib().ldpc_css_matrix(self._h, which)), owned=False)
        code._keepalive = self
        return code

    def x_stabs_binary(self):
        return self._matrix(0)

    def z_stabs_binary(self):
        return self._matrix(1)

    def x_logicals_binary(self):
        return self._matrix(2)

    def z_logicals_binary(self):
        return self._matrix(3)

    def num_x_stabs(self):
        return self.x_stabs_binary().num_checks()

    def num_z_stabs(self):
        return self.z_stabs_binary().num_checks()

    def syndrome_of(self, x_part, z_part):
        """css/mod.rs:200-205; returns (syndrome.x, syndrome.z)."""
        xa, px = _szarr(x_part)
        za, pz = _szarr(z_part)
        sx = np.zeros(max(self.num_x_stabs(), 1), dtype=np.uintp)
        sz = np.zeros(max(self.num_z_stabs(), 1), dtype=np.uintp)
        kx, kz = C.c_size_t(0), C.c_size_t(0)
        _check(lib().ldpc_css_syndrome_of(self._h, self.len(), px, xa.size, pz, za.size,
                                          sx.ctypes.data_as(C.c_void_p), C.byref(kx),
                                          sz.ctypes.data_as(C.c_void_p), C.byref(kz)))
        return [int(v) for v in sx[: kx.value]], [int(v) for v in sz[: kz.value]]

    def has_logical(self, x_part, z_part):
        xa, px = _szarr(x_part)
        za, pz = _szarr(z_part)
        r = C.c_int(0)
        _check(lib().ldpc_css_has_logical(self._h, self.len(), px, xa.size, pz, za.size, C.byref(r)))
        return bool(r.value)

    def has_stabilizer(self, x_part, z_part):
        xa, px = _szarr(x_part)
        za, pz = _szarr(z_part)
        r = C.c_int(0)
        _check(lib().ldpc_css_has_stabilizer(self._h, self.len(), px, xa.size, pz, za.size, C.byref(r)))
        return bool(r.value)

    def random_error(self, noise_model, seed, shot):
        """css/mod.rs:357-363"""
        return noise_model.sample_error_of_length(self.len(), seed, shot)


class CssErasureDecoder:
    """decoders/css_erasure.rs:7-65 (host rank algebra; small codes)."""

    def __init__(self, code):
        self.code = code

    def _bad(self, erasure):
        pos, ppos = _szarr(erasure)
        bx, bz = C.c_size_t(0), C.c_size_t(0)
        _check(lib().ldpc_css_erasure_num_bad_errors(self.code._h, self.code.len(), ppos, pos.size, C.byref(bx),
                                                     C.byref(bz)))
        return bx.value, bz.value

    def num_bad_x_errors(self, erasure):
        return self._bad(erasure)[0]

    def num_bad_z_errors(self, erasure):
        return self._bad(erasure)[1]

    def recovery_probability(self, erasure):
        pos, ppos = _szarr(erasure)
        p = C.c_double(0.0)
        _check(lib().ldpc_css_erasure_recovery_probability(self.code._h, self.code.len(), ppos, pos.size, C.byref(p)))
        return p.value


class CssDecoder:
    """CssDecoder<BpDecoder> (decoders/css.rs:4-16) as built in examples/css_decoding.rs:36-45."""

    def __init__(self, code, probability, max_iterations):
        self.code = code
        self._h = C.c_void_p()
        _check(lib().ldpc_css_bp_create(code._h, _pval(probability), int(max_iterations), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().ldpc_css_bp_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def correction_for(self, syndrome_x, syndrome_z):
        """Returns (correction.x, correction.z)."""
        sx, psx = _szarr(syndrome_x)
        sz, psz = _szarr(syndrome_z)
        n = self.code.len()
        cx = np.zeros(max(n, 1), dtype=np.uintp)
        cz = np.zeros(max(n, 1), dtype=np.uintp)
        kx, kz = C.c_size_t(0), C.c_size_t(0)
        _check(lib().ldpc_css_bp_correction_for(self._h, psx, sx.size, psz, sz.size, cx.ctypes.data_as(C.c_void_p),
                                                C.byref(kx), cz.ctypes.data_as(C.c_void_p), C.byref(kz)))
        return [int(v) for v in cx[: kx.value]], [int(v) for v in cz[: kz.value]]

    def mc_run(self, noise_probability, seed, first_shot, n_shots):
        cnt = np.zeros(N_COUNTERS, dtype=np.uint64)
        _check(lib().ldpc_css_bp_mc_run(self._h, _pval(noise_probability), int(seed), int(first_shot), int(n_shots),
                                        cnt.ctypes.data_as(C.c_void_p)))
        return cnt


def mc_run_multi(code, devices, probability, seed, first_shot, n_shots, decoder="flip", max_iterations=100):
    """One process, several GPUs (ldpc_*_mc_run_multi): `code` is a LinearCode (decoder "flip" or "bp") or a CssCode
    (decoder "css"); `devices` a list of device ordinals (a device may appear twice).  Returns the summed counters."""
    dev = (C.c_int * len(devices))(*[int(d) for d in devices])
    out = np.zeros(N_COUNTERS, dtype=np.uint64)
    po = out.ctypes.data_as(C.c_void_p)
    if decoder == "flip":
        _check(lib().ldpc_flip_mc_run_multi(code._h, len(devices), dev, _pval(probability), int(seed), int(first_shot),
                                            int(n_shots), po))
    elif decoder == "bp":
        _check(lib().ldpc_bp_mc_run_multi(code._h, len(devices), dev, _pval(probability), int(max_iterations), int(seed),
                                          int(first_shot), int(n_shots), po))
    else:
        _check(lib().ldpc_css_bp_mc_run_multi(code._h, len(devices), dev, _pval(probability), int(max_iterations),
                                              int(seed), int(first_shot), int(n_shots), po))
    return out
