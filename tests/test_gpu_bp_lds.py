"""The LDS-resident BP form (bp_lds_kernel: one workgroup per shot, messages of the shot in LDS, a thread per check /
per bit; small codes with regular checks) against the persistent and streamed forms and
the oracle.  All forms run the same arithmetic in the same order (belief_propagation.rs:150-180), so decisions,
iteration counts, counters and failure flags are compared bit for bit; LDPC_HIP_BP_PATH picks the form."""
import numpy as np
import pytest

import ldpc_amd
from ldpc_amd import (CNT_CAPPED, CNT_DETECTED, CNT_FAILURES, CNT_SHOTS, CNT_WORK, BpDecoder, CssCode, CssDecoder, LinearCode,
                      nwords)
from oracle import oracle as orc
from tests.gpu_util import assert_bp_matches_oracle, dev_u32, dev_u64, dev_u8, host_u32, host_u64, oracle_code, to_dev_u64

pytestmark = pytest.mark.gpu


def _forms(monkeypatch, fn, forms=("lds", "persistent", "stream")):
    out = {}
    for path in forms:
        monkeypatch.setenv("LDPC_HIP_BP_PATH", path)
        out[path] = fn()
        if path == "lds":
            assert ldpc_amd.last_kernel_name().startswith("bp_lds_kernel"), ldpc_amd.last_kernel_name()
        else:
            assert not ldpc_amd.last_kernel_name().startswith("bp_lds_kernel")
    monkeypatch.delenv("LDPC_HIP_BP_PATH")
    return out


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("n,m,dv,dc,p,shots,max_iter", [(1200, 900, 3, 4, 0.02, 700, 100), (1200, 900, 3, 4, 0.07, 300, 15),
                                                        (600, 300, 3, 6, 0.04, 1000, 40), (2000, 1000, 3, 6, 0.03, 200, 100),
                                                        (700, 300, 3, 7, 0.02, 400, 30), (20, 15, 3, 4, 0.1, 500, 10)])
def test_lds_correction_for_batch_equals_the_other_forms(monkeypatch, n, m, dv, dc, p, shots, max_iter):
    code = LinearCode.random_regular_code(n, m, dv, dc, seed=3)
    oc = oracle_code(code)
    syn = np.zeros((shots, nwords(m)), dtype=np.uint64)
    for i in range(shots):
        syn[i] = oc.syndrome_words(orc.sample_bsc_words(11, i, p, n))

    def run():
        dec = BpDecoder(code, p, max_iter)
        dc_, di = dev_u64((shots, nwords(n))), dev_u32((shots,))
        dec.correction_for_batch(shots, to_dev_u64(syn), dc_, di)
        return host_u64(dc_), host_u32(di)

    out = _forms(monkeypatch, run)
    assert _same(out["lds"], out["persistent"]) and _same(out["lds"], out["stream"])
    # and the oracle, on every shot, asserted: what the LDS form returns IS the reference's answer
    assert_bp_matches_oracle(code, oc, p, max_iter, syn, out["lds"][0], out["lds"][1])


def test_lds_decode_batch_and_negative_prior(monkeypatch):
    # p > 0.5: the prior is negative, the decision before the first update is all ones (belief_propagation.rs:100-113)
    code = LinearCode.random_regular_code(96, 72, 3, 4, seed=2)
    shots = 300
    for p in (0.05, 0.6):
        words = np.stack([orc.sample_bsc_words(5, i, min(p, 0.1), 96) for i in range(shots)]).reshape(shots, nwords(96))

        def run():
            dec = BpDecoder(code, p, 12)
            do = dev_u64(words.shape)
            dec.decode_batch(shots, to_dev_u64(words), do)
            return (host_u64(do),)

        out = _forms(monkeypatch, run)
        assert _same(out["lds"], out["persistent"]) and _same(out["lds"], out["stream"]), p


def test_lds_negative_prior_on_odd_and_mixed_check_degrees(monkeypatch):
    """The residual of the stop test starts as syndrome ^ H * (all ones) when the prior is negative: the parity of every
    check's degree.  Checks of odd degree (3, 7: a constant word), of mixed degree (read from the graph) and of even
    degree, from words and from given syndromes, against the other forms bit for bit."""
    from tests.gpu_util import random_sparse_rows
    rng = np.random.default_rng(4)
    codes = [LinearCode.random_regular_code(210, 90, 3, 7, seed=5), LinearCode.random_regular_code(90, 90, 3, 3, seed=6),
             LinearCode.random_regular_code(96, 72, 3, 4, seed=2),
             LinearCode.from_parity_check_matrix(130, random_sparse_rows(rng, 130, 70, 7))]
    for code in codes:
        n, m, shots = code.len(), code.num_checks(), 260
        oc = oracle_code(code)
        words = np.stack([orc.sample_bsc_words(8, i, 0.06, n) for i in range(shots)]).reshape(shots, nwords(n))
        syn = np.stack([oc.syndrome_words(words[i]) for i in range(shots)]).reshape(shots, max(nwords(m), 1))

        def run():
            dec = BpDecoder(code, 0.7, 9)
            do, di = dev_u64(words.shape), dev_u32((shots,))
            dec.decode_batch(shots, to_dev_u64(words), do, di)
            dc_, dj = dev_u64(words.shape), dev_u32((shots,))
            dec.correction_for_batch(shots, to_dev_u64(syn), dc_, dj)
            cnt = dec.mc_run(0.7, 3, 0, 400)
            return host_u64(do), host_u32(di), host_u64(dc_), host_u32(dj), cnt

        out = _forms(monkeypatch, run)
        assert _same(out["lds"], out["persistent"]) and _same(out["lds"], out["stream"]), (n, m)
        assert (out["lds"][1] == out["lds"][3]).all()    # the same runs from the word and from its syndrome


def test_lds_workgroup_sizes_change_nothing(monkeypatch):
    """bp_lds_shape gives codes of which few shots fit a CU's LDS wider workgroups (512, 768, 1024 threads); a thread
    works the checks / bits tid, tid + threads, ...: the results do not depend on the width.  The chosen widths on three
    such codes against the persistent kernel, and forced widths (LDPC_HIP_BP_LDS_THREADS) on a small code, bit for bit."""
    for (n, m, p, shots) in [(2000, 1000, 0.03, 500), (3000, 1500, 0.03, 400), (5000, 2500, 0.04, 300)]:
        code = LinearCode.random_regular_code(n, m, 3, 6, seed=4)
        oc = oracle_code(code)
        syn = np.stack([oc.syndrome_words(orc.sample_bsc_words(12, i, p, n)) for i in range(shots)])

        def run():
            dec = BpDecoder(code, p, 40)
            dc_, di = dev_u64((shots, nwords(n))), dev_u32((shots,))
            dec.correction_for_batch(shots, to_dev_u64(syn), dc_, di)
            return host_u64(dc_), host_u32(di), dec.mc_run(p, 5, 3, 2000)

        out = _forms(monkeypatch, run, forms=("lds", "persistent"))
        assert _same(out["lds"], out["persistent"]), n
        monkeypatch.delenv("LDPC_HIP_BP_PATH", raising=False)
        ldpc_amd.reload_switches()
        dflt = run()   # (and the LDS form is what a plain call takes, however large the code, as long as a shot fits)
        assert ldpc_amd.last_kernel_name().startswith("bp_lds_kernel") and _same(dflt, out["lds"])
    code = LinearCode.random_regular_code(700, 300, 3, 7, seed=3)
    monkeypatch.setenv("LDPC_HIP_BP_PATH", "lds")
    res = []
    for threads in ("64", "192", "256", "448", "1024"):
        monkeypatch.setenv("LDPC_HIP_BP_LDS_THREADS", threads)
        ldpc_amd.reload_switches()
        flags = dev_u8((3000,))
        cnt = BpDecoder(code, 0.03, 25).mc_run(0.03, 8, 0, 3000, flags)
        res.append((cnt, flags.cpu().numpy()))
    assert all(_same(res[0], r) for r in res[1:])
    # bits of four checks: the first-iteration table has 80 entries, more than a one-wave workgroup has threads (a
    # workgroup of 64 threads left the last sixteen unbuilt until round 3)
    code = LinearCode.random_regular_code(600, 400, 4, 6, seed=3)
    res = []
    for threads in ("64", "128", "256"):
        monkeypatch.setenv("LDPC_HIP_BP_LDS_THREADS", threads)
        ldpc_amd.reload_switches()
        flags = dev_u8((3000,))
        cnt = BpDecoder(code, 0.03, 25).mc_run(0.03, 8, 0, 3000, flags)
        assert ldpc_amd.last_kernel_name().startswith("bp_lds_kernel")
        res.append((cnt, flags.cpu().numpy()))
    monkeypatch.delenv("LDPC_HIP_BP_LDS_THREADS")
    monkeypatch.delenv("LDPC_HIP_BP_PATH")
    ldpc_amd.reload_switches()
    assert all(_same(res[0], r) for r in res[1:])
    want = oracle_code(code).mc_bp(0.03, 25, 8, 0, 3000, nthreads=8)
    assert [int(x) for x in res[0][0][:5]] == [int(x) for x in want[:5]]


def test_lds_hybrid_form_on_the_headline_code(monkeypatch):
    """C2 ((3,6)-regular n = 10,000: 240 KB of messages per shot) does not fit a CU's LDS; the hybrid form keeps the
    messages of the first 19,000-odd edges in LDS and the rest in the workgroup's piece of global memory.  Same
    arithmetic, same order: corrections, iteration counts, decoded words, counters and flags against the persistent
    kernel bit for bit, and it is what a plain call takes."""
    code = LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
    oc = oracle_code(code)
    for p, shots, cap in ((0.02, 600, 100), (0.06, 300, 12)):
        words = np.stack([orc.sample_bsc_words(21, i, p, 10000) for i in range(shots)])
        syn = np.stack([oc.syndrome_words(words[i]) for i in range(shots)])

        def run():
            dec = BpDecoder(code, p, cap)
            dc_, di = dev_u64(words.shape), dev_u32((shots,))
            dec.correction_for_batch(shots, to_dev_u64(syn), dc_, di)
            do, dj = dev_u64(words.shape), dev_u32((shots,))
            dec.decode_batch(shots, to_dev_u64(words), do, dj)
            flags = dev_u8((3000,))
            cnt = dec.mc_run(p, 2024, 50, 3000, flags)
            return host_u64(dc_), host_u32(di), host_u64(do), host_u32(dj), cnt, flags.cpu().numpy()

        out = _forms(monkeypatch, run, forms=("lds", "persistent"))
        assert _same(out["lds"], out["persistent"]), p
        dflt = run()
        assert ldpc_amd.last_kernel_name().startswith("bp_lds_kernel") and _same(dflt, out["lds"])
        # and the oracle, on every shot, asserted (the hybrid form's own corrections, not bp_kernel's)
        assert_bp_matches_oracle(code, oc, p, cap, syn, out["lds"][0], out["lds"][1])


def test_lds_mc_counters_flags_and_split(monkeypatch):
    code = LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)

    def run():
        dec = BpDecoder(code, 0.06, 25)
        flags = dev_u8((5000,))
        cnt = dec.mc_run(0.06, 2024, 100, 5000, flags)
        return cnt, flags.cpu().numpy()

    out = _forms(monkeypatch, run)
    assert _same(out["lds"], out["persistent"]) and _same(out["lds"], out["stream"])
    cnt, flags = out["lds"]
    assert cnt[CNT_SHOTS] == 5000 and int(flags.sum()) == cnt[CNT_FAILURES] and cnt[CNT_DETECTED] == cnt[CNT_CAPPED]
    ref = oracle_code(code).mc_bp(0.06, 25, 2024, 100, 5000, nthreads=8)
    assert abs(int(cnt[CNT_FAILURES]) - int(ref[CNT_FAILURES])) <= 2 and abs(int(cnt[CNT_WORK]) - int(ref[CNT_WORK])) <= 20
    # shot ranges add up; empty and single-shot calls
    monkeypatch.setenv("LDPC_HIP_BP_PATH", "lds")
    dec = BpDecoder(code, 0.03, 30)
    total, a, b = dec.mc_run(0.03, 7, 0, 9000), dec.mc_run(0.03, 7, 0, 4321), dec.mc_run(0.03, 7, 4321, 9000 - 4321)
    assert ((a + b) == total).all()
    assert dec.mc_run(0.03, 7, 0, 0)[CNT_SHOTS] == 0 and dec.mc_run(0.03, 7, 0, 1)[CNT_SHOTS] == 1


def test_lds_is_the_default_where_it_serves_and_css_agrees(monkeypatch):
    code = LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)
    dec = BpDecoder(code, 0.02, 100)
    dec.mc_run(0.02, 1, 0, 4096)
    assert ldpc_amd.last_kernel_name().startswith("bp_lds_kernel")
    # a code whose messages do not fit (C2: 240 KB per shot) takes the hybrid form of the kernel (bit degree 3, check
    # degree 6 only); a (4,8)-regular code of that size keeps the other forms
    big = BpDecoder(LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1), 0.02, 100)
    big.mc_run(0.02, 1, 0, 256)
    assert ldpc_amd.last_kernel_name().startswith("bp_lds_kernel")
    big = BpDecoder(LinearCode.random_regular_code(8000, 4000, 4, 8, seed=1), 0.02, 100)
    big.mc_run(0.02, 1, 0, 256)
    assert not ldpc_amd.last_kernel_name().startswith("bp_lds_kernel")
    # ... and so does a (3,6)-regular code of which LDS would hold less than half a shot (measured behind bp_kernel)
    big = BpDecoder(LinearCode.random_regular_code(16000, 8000, 3, 6, seed=1), 0.02, 100)
    big.mc_run(0.02, 1, 0, 256)
    assert not ldpc_amd.last_kernel_name().startswith("bp_lds_kernel")
    # CSS: hypergraph product of the (3,4)-regular n = 20 code with itself (the css_decoding example), X and Z halves
    css = CssCode.hypergraph_product(LinearCode.random_regular_code(20, 15, 3, 4, seed=1), LinearCode.random_regular_code(20, 15, 3, 4, seed=1))

    def run():
        return (CssDecoder(css, 0.03, 50).mc_run(0.03, 9, 0, 6000),)

    out = _forms(monkeypatch, run, forms=("lds", "persistent"))
    assert _same(out["lds"], out["persistent"])


@pytest.mark.parametrize("max_iter", [9, 30, 31, 100])
def test_lds_cuts_periodic_runs_short_and_changes_nothing(monkeypatch, max_iter):
    """Runs that never converge on the degenerate HGP code alternate between two states from about their tenth
    iteration on; the LDS form notices (bit messages equal to those of two iterations before) and stops, reporting
    what the full run reports.  Decisions and iteration counts against the persistent kernel, bit for bit, for both
    parities of the iterations left; the executed-iteration count shows that runs were in fact cut."""
    base = LinearCode.random_regular_code(20, 15, 3, 4, seed=1)
    css = CssCode.hypergraph_product(base, base)
    hx = css.x_stabs_binary()
    n, m, shots, p = hx.len(), hx.num_checks(), 3000, 0.02
    oc = oracle_code(hx)
    syn = np.zeros((shots, nwords(m)), dtype=np.uint64)
    for i in range(shots):
        syn[i] = oc.syndrome_words(orc.sample_bsc_words(3, i, p, n))

    def run():
        dec = BpDecoder(hx, p, max_iter)
        dc_, di = dev_u64((shots, nwords(n))), dev_u32((shots,))
        dec.correction_for_batch(shots, to_dev_u64(syn), dc_, di)
        return host_u64(dc_), host_u32(di)

    out = _forms(monkeypatch, run, forms=("lds", "persistent"))
    assert _same(out["lds"], out["persistent"])
    capped = int((out["lds"][1] == max_iter).sum())
    assert capped > 20, capped            # the case is there
    # Monte-Carlo counters, and the iterations actually run
    res = {}
    for path in ("lds", "persistent"):
        monkeypatch.setenv("LDPC_HIP_BP_PATH", path)
        cnt = BpDecoder(hx, p, max_iter).mc_run(p, 3, 0, shots)
        res[path] = (cnt, ldpc_amd.last_bp_executed_iterations())
    monkeypatch.delenv("LDPC_HIP_BP_PATH")
    assert (res["lds"][0] == res["persistent"][0]).all()
    assert res["persistent"][1] == res["persistent"][0][CNT_WORK]
    # (the first iteration of a shot is served from a table and not counted as run; nothing else is skipped at 9)
    cnt = res["lds"][0]
    iterated = shots - int((out["lds"][1] == 0).sum())
    if max_iter >= 30:
        assert res["lds"][1] < cnt[CNT_WORK] - iterated
    else:
        assert res["lds"][1] == cnt[CNT_WORK] - iterated


def test_lds_regular_checks_irregular_bits(monkeypatch):
    """Checks of degree 4, bits of degree 0 to beyond 8 (the bit update's general loop, no first-iteration table), NaN /
    inf arithmetic at p = 0.08: all three forms bit for bit."""
    rng = np.random.default_rng(17)
    n, m = 150, 120
    hot = rng.choice(n - 10, size=6, replace=False)          # a few bits in many checks; the last ten bits in none
    rows = []
    for c in range(m):
        pool = np.concatenate([hot, rng.choice(n - 10, size=8, replace=False)]) if c % 2 == 0 else rng.choice(n - 10, size=12, replace=False)
        rows.append(sorted(int(x) for x in rng.choice(np.unique(pool), size=4, replace=False)))
    code = LinearCode.from_parity_check_matrix(n, rows)
    degs = np.bincount(np.concatenate(rows), minlength=n)
    assert degs.max() > 8 and degs.min() == 0
    shots = 400
    words = np.stack([orc.sample_bsc_words(5, i, 0.08, n) for i in range(shots)]).reshape(shots, nwords(n))

    def run():
        dec = BpDecoder(code, 0.08, 20)
        do = dev_u64(words.shape)
        dec.decode_batch(shots, to_dev_u64(words), do)
        return (host_u64(do),)

    out = _forms(monkeypatch, run)
    assert _same(out["lds"], out["persistent"]) and _same(out["lds"], out["stream"])
    # max_iterations = 0: nothing is updated, the first decision (all zeros at a positive prior) comes back
    monkeypatch.setenv("LDPC_HIP_BP_PATH", "lds")
    do = dev_u64(words.shape)
    BpDecoder(code, 0.08, 0).decode_batch(shots, to_dev_u64(words), do)
    assert (host_u64(do) == words).all()


def test_lds_checks_of_any_degree(monkeypatch):
    """Irregular rows (degree 1 to 12, empty rows, degree-0 bits), the repetition code (check degree 2) and the
    Hamming code: the LDS form's general check row against the other forms, bit for bit, at p = 0.08 (NaN / inf
    arithmetic on the shots that do not converge)."""
    from tests.gpu_util import random_sparse_rows
    rng = np.random.default_rng(9)
    codes = [LinearCode.from_parity_check_matrix(n, random_sparse_rows(rng, n, m, maxdeg))
             for (n, m, maxdeg) in [(7, 3, 4), (65, 40, 6), (130, 90, 12), (300, 200, 5)]]
    codes += [LinearCode.repetition_code(9), LinearCode.hamming_code()]
    for code in codes:
        n, shots = code.len(), 200
        words = np.stack([orc.sample_bsc_words(5, i, 0.08, n) for i in range(shots)]).reshape(shots, max(nwords(n), 1))

        def run():
            dec = BpDecoder(code, 0.08, 12)
            do = dev_u64(words.shape)
            dec.decode_batch(shots, to_dev_u64(words), do)
            cnt = dec.mc_run(0.08, 4, 0, 500)
            return host_u64(do), cnt

        out = _forms(monkeypatch, run)
        assert _same(out["lds"], out["persistent"]) and _same(out["lds"], out["stream"]), n


def test_lds_code_with_hardly_any_edges(monkeypatch):
    """Most bits in no check: the message array (8 bytes per edge) is shorter than a list of every error position, so
    the set-up keeps the scan of the word (the list borrows that array; round 3).  Checks of degree 3 and bits of
    degree at most 2 take the form with the first-iteration table, the one that lists: all forms bit for bit, and the
    oracle's decisions."""
    n = 640
    rows = [[3, 200, 411], [3, 17, 599], [64, 200, 630]]
    code = LinearCode.from_parity_check_matrix(n, rows)
    shots = 300
    words = np.stack([orc.sample_bsc_words(11, i, 0.3, n) for i in range(shots)]).reshape(shots, nwords(n))

    def run():
        dec = BpDecoder(code, 0.3, 10)
        do = dev_u64(words.shape)
        dec.decode_batch(shots, to_dev_u64(words), do)
        cnt = dec.mc_run(0.3, 4, 0, 500)
        return host_u64(do), cnt

    out = _forms(monkeypatch, run)
    assert _same(out["lds"], out["persistent"]) and _same(out["lds"], out["stream"])
    oc = oracle_code(code)
    want = np.stack([oc.bp_decode_words(0.3, 10, words[i])[0] for i in range(shots)])
    assert (out["lds"][0] == want).all(axis=1).mean() >= 0.9999
    cnt = oc.mc_bp(0.3, 10, 4, 0, 500)
    assert [int(x) for x in out["lds"][1][:5]] == [int(x) for x in cnt[:5]]
