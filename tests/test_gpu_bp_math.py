"""Accuracy and edge behaviour of the BP kernel's own tanh(x/2) and log1p (csrc/bp_math.hpp),
measured on the MI355X against mpmath at 60 digits.  Bar: tanh(x/2) <= 0.6 ulp (0.7 for |x| < 0.5), log1p <= 0.85 ulp
(glibc's tanh reaches 1.15 ulp, OCML's log1p 0.62: scripts/math_ulps.py) and exact
IEEE results at the edges the un-clamped BP arithmetic reaches."""
import ctypes as C
import math

import mpmath
import numpy as np
import pytest

import ldpc_amd

pytestmark = pytest.mark.gpu


def _run(kind, x, n=None):
    """n results; the two-operand kinds (4, 6) read x[t] and x[n + t], so x holds 2n values for them."""
    import torch
    L = ldpc_amd.lib()
    fn = L.ldpc_internal_bp_math
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
    d_in = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to("cuda")
    n = d_in.numel() if n is None else n
    assert d_in.numel() >= (2 * n if kind in (4, 6) else n)
    d_out = torch.empty(2 * n if kind in (7, 8, 9, 10) else n, dtype=torch.float64, device="cuda")   # (the batch kinds also return their counts)
    assert fn(kind, n, d_in.data_ptr(), d_out.data_ptr()) == 0
    return d_out.cpu().numpy()


def _ulp_err(got, ref_fn, xs):
    mpmath.mp.dps = 60
    worst = 0.0
    for g, x in zip(got, xs):
        r = ref_fn(mpmath.mpf(float(x)))
        rf = float(r)
        if rf == 0.0:
            assert g == 0.0
            continue
        ulp = math.ulp(rf)
        worst = max(worst, abs(float((mpmath.mpf(float(g)) - r) / ulp)))
    return worst


def test_tanh_half_accuracy():
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-80, 80, 3000), rng.uniform(-1, 1, 3000), rng.uniform(-0.51, -0.49, 200),
                         rng.uniform(0.49, 0.51, 200), 10.0 ** rng.uniform(-300, 0, 500), rng.uniform(0.0035, 0.0045, 300), rng.uniform(36, 39, 300),
                         np.array([3.8918202981106265, 7.783640596221253, 0.5, -0.5, 1e-320, 745.0, -745.0])])
    got = _run(0, xs)
    big = np.abs(xs) >= 0.5
    assert _ulp_err(got[big], lambda v: mpmath.tanh(v / 2), xs[big]) <= 0.6
    assert _ulp_err(got[~big], lambda v: mpmath.tanh(v / 2), xs[~big]) <= 0.7
    assert (np.abs(got) <= 1.0).all()


def test_tanh_half_edges():
    xs = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 100.0, -100.0, 1e300, -1e300])
    got = _run(0, xs)
    assert got[0] == 0.0 and not np.signbit(got[0])
    assert got[1] == 0.0 and np.signbit(got[1])
    assert got[2] == 1.0 and got[3] == -1.0 and np.isnan(got[4])
    assert got[5] == 1.0 and got[6] == -1.0 and got[7] == 1.0 and got[8] == -1.0     # saturates exactly, as libm does


def test_log1p_accuracy():
    rng = np.random.default_rng(2)
    xs = np.concatenate([rng.uniform(-0.999999, 5, 4000), 10.0 ** rng.uniform(-300, 300, 2000),
                         -(10.0 ** rng.uniform(-300, -1e-9, 1000)), rng.uniform(-0.30, -0.28, 300),
                         rng.uniform(0.40, 0.43, 300), -1 + 10.0 ** rng.uniform(-16, -1, 500),
                         np.array([0.41421356237309515, -0.2928932188134524, 2.0 ** 53, 2.0 ** 54, 1e-320])])
    got = _run(1, xs)
    assert _ulp_err(got, mpmath.log1p, xs) <= 0.85


def test_log1p_edges():
    xs = np.array([0.0, -0.0, -1.0, -1.0000000000000002, -2.0, -np.inf, np.inf, np.nan, 1.7976931348623157e308])
    got = _run(1, xs)
    assert got[0] == 0.0 and not np.signbit(got[0])
    assert got[1] == 0.0 and np.signbit(got[1])
    assert got[2] == -np.inf
    assert np.isnan(got[3]) and np.isnan(got[4]) and np.isnan(got[5])
    assert got[6] == np.inf and np.isnan(got[7])
    assert abs(got[8] - 709.782712893384) < 1e-9


def test_fast_division_is_ieee():
    """bp_div_core (reciprocal + Newton step + residual correction) against the IEEE quotient on the operand
    ranges of its call sites; bp_div_prod's guard on zero, denormal and NaN operands."""
    rng = np.random.default_rng(7)
    n = 1 << 20
    y = np.concatenate([rng.uniform(-1, 1, n // 2), 1 - 2.0 ** rng.uniform(-53, -1, n // 2)])
    f = rng.uniform(-0.3, 0.42, n)
    for a, b in ((rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)), (f, 2.0 + f), (2 * y, 1 - y)):
        got = _run(4, np.concatenate([a, b]), n)
        assert (got != a / b).mean() <= 1e-6
    a = np.full(64, 0.25)
    b = np.full(64, 0.5)
    a[:8] = [0.0, 0.0, 1e-320, 1e-300, 0.5, np.nan, 1e-310, -0.0]
    b[:8] = [0.0, 1.0, 1e-310, 1e-300, 0.0, 1.0, 1e-310, 0.0]
    got = _run(6, np.concatenate([a, b]), 64)
    with np.errstate(all="ignore"):
        ref = a / b
    assert all((g == r) or (g != g and r != r) for g, r in zip(got, ref))


def test_ref_atanh_matches_formula():
    """bp_ref_atanh(x) = 0.5 * log1p(2x / (1 - x)) incl. x = 1 (+inf), x = 0, NaN."""
    rng = np.random.default_rng(8)
    xs = np.concatenate([rng.uniform(-1, 1, 4000), 1 - 2.0 ** rng.uniform(-53, -1, 1000), [1.0, 0.0, -0.0, -1.0, np.nan]])
    got = _run(5, xs)
    # (the fast division returns +0 for a -0 dividend: the sign of a zero message reaches no other value or decision)
    assert got[-5] == np.inf and got[-4] == 0.0 and got[-3] == 0.0 and got[-2] == -np.inf and np.isnan(got[-1])
    mpmath.mp.dps = 60
    for g, x in zip(got[:-5], xs[:-5]):
        q = (2.0 * x) / (1.0 - x)                       # the reference rounds this quotient before ln_1p
        r = mpmath.log1p(mpmath.mpf(float(q))) / 2
        assert abs(float((mpmath.mpf(float(g)) - r) / math.ulp(float(r)))) <= 0.85


@pytest.mark.parametrize("kind", [7, 8, 9, 10])
def test_tanh_batch_equals_the_single_form(kind):
    """bp_tanh_half_n (the tanh of a bit's messages as one batch, constants built once) against bp_tanh_half, bit for
    bit, for every count 0..8 of a batch of eight -- the same count in the whole wave (7, 9) or a different one per lane
    (8, 10), with scalar-built (7, 8) or plain (9, 10) constants -- NaN, infinities and zeros among the values.  Round 2
    dropped a batch for bits of any degree up to eight after a wrong result in one irregular code; rebuilt in round 3
    (-DBPL_TANH_BATCH_ANY) it is bit-identical on every code of the suite (scripts/tanh_batch_repro.py), and this test
    pins the routine itself: elements below the count equal the single form, elements from the count on are untouched."""
    rng = np.random.default_rng(kind)
    n = 8 * 4096
    xs = np.concatenate([rng.uniform(-80, 80, n - 64), 10.0 ** rng.uniform(-300, 2, 48), np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 745.0,
                         -745.0, 1e-320, 0.5, -0.5, 38.0, -38.0, 64.0, -64.0, 1e300, -1e300])])
    rng.shuffle(xs)
    single = _run(0, xs)
    out = _run(kind, xs)
    got, cnt = out[:n], out[n:].astype(np.int64)
    j = np.arange(n) % 8
    assert set(np.unique(cnt)) == set(range(9))
    done = j < cnt
    assert np.array_equal(got[done].view(np.uint64), single[done].view(np.uint64))
    assert (got[~done] == -7.0).all()
