"""Wide BP parity runs on the MI355X (the comparisons that used to live in scripts/bp_stress.py only).

* configs[1] (C2) on 10 240 shots against the multi-threaded oracle: decisions >= 99.99 %, iteration counts,
  posteriors; shots that differ are followed iteration by iteration to where the messages first part.
* check -> bit messages (belief_propagation.rs:156-163) at rtol 1e-5 after iteration 1, 2, 3 and at stop, shots
  stopped by the cap included.
* configs[3] end points p = 0.005 and 0.05 on the C2 code, both decoders, counters against the oracle's.
* irregular matrices, NaN / inf arithmetic: decisions are compared wherever the messages still agree, iteration
  by iteration, instead of accepting a fixed share of differing final decisions."""
import numpy as np
import pytest

from ldpc_amd import (CNT_CAPPED, CNT_DETECTED, CNT_FAILURES, CNT_SHOTS, CNT_WORK, BinarySymmetricChannel, BpDecoder,
                      FlipDecoder, LinearCode, nwords)
from oracle import oracle as orc
from tests.gpu_util import dev_f64, dev_u32, dev_u64, host_u32, host_u64, oracle_code, random_sparse_rows, to_dev_u64

pytestmark = pytest.mark.gpu

SEED = 2024
LLR_RTOL = 1e-5
THREADS = 16


def _syndromes(code, oc, p, shots, seed=SEED, first=0):
    n, m = code.len(), code.num_checks()
    syn = np.zeros((shots, max(nwords(m), 1)), dtype=np.uint64)
    for i in range(shots):
        syn[i, : nwords(m)] = oc.syndrome_words(orc.sample_bsc_words(seed, first + i, p, n))
    return syn


def _gpu(code, p, cap, syn, messages=False):
    n, shots = code.len(), syn.shape[0]
    dec = BpDecoder(code, p, cap)
    dc, di, dp = dev_u64((shots, max(nwords(n), 1))), dev_u32((shots,)), dev_f64((shots, max(n, 1)))
    dm = dev_f64((shots, max(code.num_edges(), 1))) if messages else None
    dec.correction_for_batch(shots, to_dev_u64(syn), dc, di, dp, dm)
    return host_u64(dc), host_u32(di), dp.cpu().numpy(), (dm.cpu().numpy() if messages else None)


def _agree(a, b, rtol=LLR_RTOL):
    """rows of a, b agree: same non-finite pattern (NaN = NaN, inf = inf with sign), finite values within rtol"""
    fa, fb = np.isfinite(a), np.isfinite(b)
    same_kind = (fa == fb) & (np.isnan(a) == np.isnan(b)) & (fa | np.isnan(a) | (a == b))
    close = np.isclose(np.where(fa, a, 0.0), np.where(fb, b, 0.0), rtol=rtol, atol=0.0)
    return (same_kind & (close | ~fa)).all(axis=1)


@pytest.mark.parametrize("form", ["default", "persistent"])
def test_bp_c2_wide(form, monkeypatch):
    if form == "persistent":
        monkeypatch.setenv("LDPC_HIP_BP_PATH", "persistent")
    n, m, p, shots = 10000, 5000, 0.02, 10240
    code = LinearCode.random_regular_code(n, m, 3, 6, seed=1)
    oc = oracle_code(code)
    de, ds = dev_u64((shots, nwords(n))), dev_u64((shots, nwords(m)))
    BinarySymmetricChannel(p).sample_batch(n, SEED, 0, shots, de)
    code.syndrome_batch(shots, de, ds)
    syn = host_u64(ds)
    dec = BpDecoder(code, p, 100)
    dc, di, dp = dev_u64((shots, nwords(n))), dev_u32((shots,)), dev_f64((shots, n))
    dec.correction_for_batch(shots, ds, dc, di, dp)
    gc, gi, gp = host_u64(dc), host_u32(di), dp.cpu().numpy()
    rc, ri, rp, _ = oc.bp_correction_for_batch(p, 100, syn, want_post=True, nthreads=THREADS)
    same = (gc == rc).all(axis=1)
    assert same.mean() >= 0.9999, int((~same).sum())
    assert (gi[same] == ri[same]).all()
    assert _agree(gp[same], rp[same]).all()
    # a shot that differs: the messages must agree through its first iterations (the split is rounding noise
    # amplified over many iterations of a shot that does not converge, SURVEY.md section 3.2, not arithmetic)
    for i in np.nonzero(~same)[0][:4]:
        for cap in (1, 2, 3):
            _, git, _, gm = _gpu(code, p, cap, syn[i : i + 1], messages=True)
            _, rit, _, rm = oc.bp_correction_for_batch(p, cap, syn[i : i + 1], want_messages=True)
            assert git[0] == rit[0] and _agree(gm, rm).all(), (int(i), cap)


@pytest.mark.parametrize("n,m,dv,dc,p,shots,caps", [(1200, 900, 3, 4, 0.02, 300, (1, 2, 3, 100)),
                                                    (1200, 900, 3, 4, 0.07, 300, (1, 2, 3, 6)),      # most shots stop by the cap
                                                    (10000, 5000, 3, 6, 0.02, 96, (1, 2, 100)),
                                                    (10000, 5000, 3, 6, 0.05, 96, (1, 2, 3))])
@pytest.mark.parametrize("form", ["default", "persistent"])
def test_bp_check_messages_match_oracle(form, monkeypatch, n, m, dv, dc, p, shots, caps):
    if form == "persistent":
        monkeypatch.setenv("LDPC_HIP_BP_PATH", "persistent")
    code = LinearCode.random_regular_code(n, m, dv, dc, seed=1)
    oc = oracle_code(code)
    syn = _syndromes(code, oc, p, shots)
    for cap in caps:
        gc, gi, gp, gm = _gpu(code, p, cap, syn, messages=True)
        rc, ri, rp, rm = oc.bp_correction_for_batch(p, cap, syn, want_post=True, want_messages=True, nthreads=THREADS)
        ok = _agree(gm, rm)
        # through the first iterations every shot's messages agree; later a non-converging shot may drift
        need = 1.0 if cap <= 3 else 0.99
        assert ok.mean() >= need, (cap, int((~ok).sum()))
        assert (gi[ok] == ri[ok]).all() and (gc[ok] == rc[ok]).all(), cap
        assert _agree(gp[ok], rp[ok]).all(), cap
        assert (gi <= cap).all()


@pytest.mark.parametrize("p", [0.005, 0.05])
def test_p_sweep_end_points_flip(p):
    # configs[3] on the C2 code: exact counters (integer decoder)
    code = LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
    oc = oracle_code(code)
    shots = 6000
    cnt = FlipDecoder(code).mc_run(p, SEED, 0, shots)
    ref = oc.mc_flip(p, SEED, 0, shots, nthreads=THREADS)
    assert (cnt == ref).all(), (cnt, ref)


@pytest.mark.parametrize("p,shots", [(0.005, 3000), (0.05, 1500)])
def test_p_sweep_end_points_bp(p, shots):
    code = LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
    oc = oracle_code(code)
    cnt = BpDecoder(code, p, 100).mc_run(p, SEED, 0, shots)
    ref = oc.mc_bp(p, 100, SEED, 0, shots, nthreads=THREADS)
    assert cnt[CNT_SHOTS] == shots
    tol = max(2, int(1e-4 * shots))
    assert abs(int(cnt[CNT_FAILURES]) - int(ref[CNT_FAILURES])) <= tol, (cnt, ref)
    assert abs(int(cnt[CNT_DETECTED]) - int(ref[CNT_DETECTED])) <= tol
    # iterations: equal on every shot that decodes alike; a differing shot can add at most the cap
    assert abs(int(cnt[CNT_WORK]) - int(ref[CNT_WORK])) <= 100 * tol, (cnt, ref)


@pytest.mark.parametrize("form", ["default", "persistent"])
def test_bp_irregular_iteration_by_iteration(form, monkeypatch):
    """Irregular rows up to degree 12, empty rows, degree-0 bits, p = 0.08: NaN / inf arithmetic on the shots that do
    not converge.  For every cap k = 1..12: wherever the messages of the two implementations still agree, so do the
    decisions and the iteration counts; and they agree on every shot through the first iterations."""
    if form == "persistent":
        monkeypatch.setenv("LDPC_HIP_BP_PATH", "persistent")
    rng = np.random.default_rng(5)
    for (n, m, maxdeg) in [(7, 3, 4), (65, 40, 6), (130, 90, 12), (40, 30, 3), (200, 150, 9)]:
        rows = random_sparse_rows(rng, n, m, maxdeg)
        code = LinearCode.from_parity_check_matrix(n, rows)
        oc = oracle_code(code)
        shots = 140
        syn = _syndromes(code, oc, 0.08, shots, seed=1)
        share = []
        for cap in range(1, 13):
            gc, gi, gp, gm = _gpu(code, 0.08, cap, syn, messages=True)
            rc, ri, rp, rm = oc.bp_correction_for_batch(0.08, cap, syn, want_post=True, want_messages=True, nthreads=THREADS)
            ok = _agree(gm, rm)
            share.append(ok.mean())
            assert (gi[ok] == ri[ok]).all(), (n, cap)
            differ = ok & ~(gc == rc).all(axis=1)
            for i in np.nonzero(differ)[0]:
                # equal messages, different decision: only a posterior that is zero up to rounding may do that
                bits = np.nonzero(np.unpackbits((gc[i] ^ rc[i]).view(np.uint8), bitorder="little")[:n])[0]
                assert (np.abs(rp[i, bits]) < 1e-9).all(), (n, cap, int(i))
        assert share[0] == 1.0 and share[1] == 1.0, (n, share)
        assert share[-1] >= 0.97, (n, share)
