"""Belief propagation on the MI355X against the CPU oracle.

Float path, so the bar is north_star's: decisions equal on >= 99.99 % of shots,
posterior LLRs within 1e-5 relative, logical error rate inside the oracle's
binomial 95 % interval.  (On the converging shots below the observed agreement
is exact; the tolerances are what the tests assert.)"""
import numpy as np
import pytest

from ldpc_amd import (CNT_CAPPED, CNT_FAILURES, CNT_SHOTS, CNT_WORK, BpDecoder, LinearCode, nwords)
from oracle import oracle as orc
from tests.gpu_util import dev_f64, dev_u32, dev_u64, dev_u8, host_u32, host_u64, oracle_code, random_sparse_rows, to_dev_u64

pytestmark = pytest.mark.gpu

SEED = 2024


@pytest.fixture(autouse=True, params=["default", "persistent"])
def bp_form(request, monkeypatch):
    """Small calls on codes with E >= 3000 take the streamed kernels by default (capi.cpp run_bp); every test here
    runs both with that choice and with the persistent kernel forced, each against the oracle."""
    if request.param == "persistent":
        monkeypatch.setenv("LDPC_HIP_BP_PATH", "persistent")
    return request.param


LLR_RTOL = 1e-5
DECISION_MATCH = 0.9999


def _compare_batch(code, p, max_iter, syndromes, want_exact_iters=True):
    """syndromes: [shots][Wm] u64.  Returns fraction of shots with equal corrections."""
    n, m = code.len(), code.num_checks()
    oc = oracle_code(code)
    dec = BpDecoder(code, p, max_iter)
    shots = syndromes.shape[0]
    dc, di, dp = dev_u64((shots, max(nwords(n), 1))), dev_u32((shots,)), dev_f64((shots, max(n, 1)))
    dec.correction_for_batch(shots, to_dev_u64(syndromes), dc, di, dp)
    gc, gi, gp = host_u64(dc), host_u32(di), dp.cpu().numpy()
    same = 0
    for i in range(shots):
        c, it, post = oc.bp_correction_for(p, max_iter, syndromes[i, : nwords(m)])
        eq = (gc[i, : nwords(n)] == c).all()
        same += bool(eq)
        if eq and want_exact_iters:
            assert gi[i] == it
            fin = np.isfinite(post)
            assert (np.isfinite(gp[i, :n]) == fin).all()
            assert np.allclose(gp[i, :n][fin], post[fin], rtol=LLR_RTOL, atol=0)
    return same / max(shots, 1)


def test_bp_golden_c1():
    g = np.load("tests/golden/c1_vectors.npz")
    code = LinearCode.from_csr(1200, g["h_ptr"], g["h_idx"])
    dec = BpDecoder(code, 0.02, 100)
    shots = g["syndromes"].shape[0]
    dc, di = dev_u64(g["bp_corr"].shape), dev_u32((shots,))
    dec.correction_for_batch(shots, to_dev_u64(g["syndromes"]), dc, di)
    match = (host_u64(dc) == g["bp_corr"]).all(axis=1)
    assert match.mean() >= DECISION_MATCH
    assert (host_u32(di)[match] == g["bp_iters"][match]).all()
    # decode() = message ^ correction (belief_propagation.rs:15-19)
    do = dev_u64(g["errors"].shape)
    dec.decode_batch(shots, to_dev_u64(g["errors"]), do)
    assert ((host_u64(do) == (g["errors"] ^ g["bp_corr"])).all(axis=1)).mean() >= DECISION_MATCH


@pytest.mark.parametrize("n,m,dv,dc,p,shots", [(1200, 900, 3, 4, 0.02, 192), (1200, 900, 3, 4, 0.06, 128),
                                               (10000, 5000, 3, 6, 0.02, 64), (16, 12, 3, 4, 0.1, 128)])
def test_bp_matches_oracle(n, m, dv, dc, p, shots):
    code = LinearCode.random_regular_code(n, m, dv, dc, seed=1)
    oc = oracle_code(code)
    syn = np.zeros((shots, nwords(m)), dtype=np.uint64)
    for i in range(shots):
        syn[i] = oc.syndrome_words(orc.sample_bsc_words(SEED, i, p, n))
    assert _compare_batch(code, p, 100, syn) >= DECISION_MATCH


def test_bp_irregular_and_edge_cases():
    rng = np.random.default_rng(5)
    for (n, m, maxdeg) in [(1, 0, 1), (7, 3, 4), (65, 40, 6), (130, 90, 12), (40, 30, 3)]:
        rows = random_sparse_rows(rng, n, m, maxdeg)     # includes empty rows, degree-0 bits, degree > 8 rows
        code = LinearCode.from_parity_check_matrix(n, rows)
        oc = oracle_code(code)
        shots = 70
        syn = np.zeros((shots, max(nwords(m), 1)), dtype=np.uint64)
        for i in range(shots):
            syn[i, : nwords(m)] = oc.syndrome_words(orc.sample_bsc_words(1, i, 0.08, n))
        # max_iter small: non-converging shots stop by the cap, NaN/inf arithmetic included
        assert _compare_batch(code, 0.08, 12, syn) >= 0.97


def test_bp_extreme_priors():
    code = LinearCode.hamming_code()
    oc = oracle_code(code)
    syn = np.array([[i] for i in range(8)], dtype=np.uint64)
    for p in (0.0, 1.0, 0.5, 0.9, 1e-12):
        dec = BpDecoder(code, p, 5)
        dc, di = dev_u64((8, 1)), dev_u32((8,))
        dec.correction_for_batch(8, to_dev_u64(syn), dc, di)
        for i in range(8):
            c, it, _ = oc.bp_correction_for(p, 5, syn[i])
            assert host_u64(dc)[i, 0] == c[0] and host_u32(di)[i] == it, (p, i)


def test_bp_iteration_caps_zero_and_one():
    """max_iterations = 0: the loop of belief_propagation.rs:26-29 never runs and the decision of the priors is
    returned; 1: exactly one update.  Hamming code (persistent kernel) and C1 (a call this small runs the streamed
    kernels unless forced), against the oracle."""
    rng = np.random.default_rng(3)
    for code, shots in ((LinearCode.hamming_code(), 8), (LinearCode.random_regular_code(1200, 900, 3, 4, seed=1), 40)):
        n, m = code.len(), code.num_checks()
        oc = oracle_code(code)
        syn = np.stack([oc.syndrome_words(orc.sample_bsc_words(5, i, 0.03 if n > 7 else 0.2, n)) for i in range(shots)])
        syn = syn.reshape(shots, max(nwords(m), 1))
        for p in (0.03, 0.7):
            for cap in (0, 1):
                dec = BpDecoder(code, p, cap)
                dc, di = dev_u64((shots, nwords(n))), dev_u32((shots,))
                dec.correction_for_batch(shots, to_dev_u64(syn), dc, di)
                gc, gi = host_u64(dc), host_u32(di)
                for i in range(shots):
                    c, it, _ = oc.bp_correction_for(p, cap, syn[i])
                    assert (gc[i] == c).all() and gi[i] == it, (n, p, cap, i)
                assert int(gi.max()) <= cap


def test_bp_mc_counters_c1():
    g = np.load("tests/golden/c1_vectors.npz")
    code = LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)
    dec = BpDecoder(code, 0.02, 100)
    flags = dev_u8((2000,))
    cnt = dec.mc_run(0.02, SEED, 0, 2000, flags)
    ref = g["c1_bp_counters_2000"]
    assert cnt[CNT_SHOTS] == 2000
    assert cnt[CNT_FAILURES] == ref[CNT_FAILURES] and cnt[CNT_CAPPED] == ref[CNT_CAPPED]
    assert abs(int(cnt[CNT_WORK]) - int(ref[CNT_WORK])) <= 2     # iterations: float path
    assert int(flags.sum()) == cnt[CNT_FAILURES]


def test_bp_mc_logical_error_rate_within_oracle_ci():
    # harder point so that failures exist: (3,4) n=1200 at p=0.08, 20 iterations
    code = LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)
    oc = oracle_code(code)
    shots = 1500
    ref = oc.mc_bp(0.08, 20, SEED, 0, shots, nthreads=8)
    cnt = BpDecoder(code, 0.08, 20).mc_run(0.08, SEED, 0, shots)
    pr = ref[CNT_FAILURES] / shots
    half = 1.96 * (max(pr * (1 - pr), 1e-4) / shots) ** 0.5
    assert abs(cnt[CNT_FAILURES] / shots - pr) <= half
    # shot-by-shot the agreement is much tighter than the interval
    assert abs(int(cnt[CNT_FAILURES]) - int(ref[CNT_FAILURES])) <= max(2, int(1e-4 * shots))


def test_bp_refill_and_split_invariance():
    # more shots than one pass of the slots: harvest/refill path; counters add over a split range
    code = LinearCode.random_regular_code(96, 72, 3, 4, seed=2)
    dec = BpDecoder(code, 0.05, 30)
    total = dec.mc_run(0.05, 7, 0, 20000)
    a = dec.mc_run(0.05, 7, 0, 12345)
    b = dec.mc_run(0.05, 7, 12345, 20000 - 12345)
    assert ((a + b) == total).all()
    oc = oracle_code(code)
    ref = oc.mc_bp(0.05, 30, 7, 0, 20000, nthreads=8)
    assert abs(int(total[CNT_FAILURES]) - int(ref[CNT_FAILURES])) <= 2
    assert abs(int(total[CNT_WORK]) - int(ref[CNT_WORK])) <= 20


def test_bp_full_size_properties():
    # C2 shape, 2^15 shots, properties that need no oracle: where BP stops before the cap the correction
    # reproduces the syndrome (H c = s, belief_propagation.rs:26-27); decode(e) = e ^ correction_for(H e);
    # Monte-Carlo failures = shots whose correction differs from the error; counters add over a split range
    n, m, shots, p, cap = 10000, 5000, 1 << 15, 0.03, 30
    code = LinearCode.random_regular_code(n, m, 3, 6, seed=1)
    dec = BpDecoder(code, p, cap)
    from ldpc_amd import BinarySymmetricChannel
    de, ds = dev_u64((shots, nwords(n))), dev_u64((shots, nwords(m)))
    BinarySymmetricChannel(p).sample_batch(n, SEED, 0, shots, de)
    code.syndrome_batch(shots, de, ds)
    dc, di = dev_u64((shots, nwords(n))), dev_u32((shots,))
    dec.correction_for_batch(shots, ds, dc, di)
    chk = dev_u64((shots, nwords(m)))
    code.syndrome_batch(shots, dc, chk)
    stopped = di < cap
    assert int(stopped.sum()) > shots * 0.99
    assert bool((chk == ds)[stopped].all())
    do = dev_u64((shots, nwords(n)))
    dec.decode_batch(shots, de, do)
    assert bool((do == (de ^ dc)).all())
    flags = dev_u8((shots,))
    cnt = dec.mc_run(p, SEED, 0, shots, flags)
    assert int((do != 0).any(dim=1).sum()) == cnt[CNT_FAILURES] == int(flags.sum())
    assert int(di.sum()) == cnt[CNT_WORK] and int((~stopped).sum()) >= cnt[CNT_CAPPED]
    a = dec.mc_run(p, SEED, 0, 12345)
    b = dec.mc_run(p, SEED, 12345, shots - 12345)
    assert ((a + b) == cnt).all()


def test_bp_rare_failures_are_the_oracles():
    # C2 at p = 0.02 fails about once in 10^5 shots.  The shots the device run flags as failures are decoded by
    # the oracle one by one: each must fail there too (same correction or at least a non-zero residual with the
    # same stop), and a sample of unflagged shots must succeed.
    n, m, shots, p = 10000, 5000, 1 << 19, 0.02
    code = LinearCode.random_regular_code(n, m, 3, 6, seed=1)
    oc = oracle_code(code)
    dec = BpDecoder(code, p, 100)
    flags = dev_u8((shots,))
    cnt = dec.mc_run(p, 1, 0, shots, flags)
    failing = np.nonzero(flags.cpu().numpy())[0]
    assert len(failing) == cnt[CNT_FAILURES] and 0 < len(failing) < 50
    for i in failing:
        e = orc.sample_bsc_words(1, int(i), p, n)
        out, _ = oc.bp_decode_words(p, 100, e)
        assert out.any(), i
    rng = np.random.default_rng(0)
    for i in rng.choice(shots, 40, replace=False):
        if i in failing:
            continue
        e = orc.sample_bsc_words(1, int(i), p, n)
        out, _ = oc.bp_decode_words(p, 100, e)
        assert not out.any(), i
