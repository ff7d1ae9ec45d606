"""The streamed BP form (one grid-wide kernel per phase; used when a 64-shot group's messages are so large that
memory allows few groups, C5) against the persistent kernel and the oracle.  Both forms run the same arithmetic
in the same order, so their outputs are compared bit for bit; LDPC_HIP_BP_PATH forces the form on small codes."""
import numpy as np
import pytest

from ldpc_amd import CNT_FAILURES, CNT_SHOTS, CNT_WORK, BpDecoder, LinearCode, nwords
from oracle import oracle as orc
from tests.gpu_util import dev_f64, dev_u32, dev_u64, dev_u8, host_u32, host_u64, oracle_code, random_sparse_rows, to_dev_u64

pytestmark = pytest.mark.gpu


def _both(monkeypatch, fn):
    """persistent kernel, streamed form with every shot loaded at once (short control), and streamed form held to
    two groups so that slots are refilled from the shot queue (the C5 regime); the last two must agree."""
    out = {}
    for path in ("persistent", "stream"):
        monkeypatch.setenv("LDPC_HIP_BP_PATH", path)
        out[path] = fn()
    monkeypatch.setenv("LDPC_HIP_BP_GROUPS", "2")
    refilled = fn()
    monkeypatch.delenv("LDPC_HIP_BP_GROUPS")
    monkeypatch.delenv("LDPC_HIP_BP_PATH")
    assert _same(out["stream"], refilled)
    return out["persistent"], out["stream"]


def _same(a, b):
    return all(np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("n,m,dv,dc,p,shots,max_iter", [(1200, 900, 3, 4, 0.02, 300, 100), (1200, 900, 3, 4, 0.07, 200, 15),
                                                        (10000, 5000, 3, 6, 0.02, 130, 100), (600, 300, 3, 6, 0.04, 1000, 40)])
def test_stream_correction_for_equals_persistent(monkeypatch, n, m, dv, dc, p, shots, max_iter):
    code = LinearCode.random_regular_code(n, m, dv, dc, seed=3)
    oc = oracle_code(code)
    syn = np.zeros((shots, nwords(m)), dtype=np.uint64)
    for i in range(shots):
        syn[i] = oc.syndrome_words(orc.sample_bsc_words(11, i, p, n))

    def run():
        dec = BpDecoder(code, p, max_iter)
        dc_, di, dp = dev_u64((shots, nwords(n))), dev_u32((shots,)), dev_f64((shots, n))
        dec.correction_for_batch(shots, to_dev_u64(syn), dc_, di, dp)
        return host_u64(dc_), host_u32(di), dp.cpu().numpy()

    pers, strm = _both(monkeypatch, run)
    assert _same(pers, strm)
    # and the oracle on a sample
    for i in range(0, shots, 37):
        c, it, _ = oc.bp_correction_for(p, max_iter, syn[i])
        if (strm[0][i] == c).all():
            assert strm[1][i] == it


def test_stream_irregular_codes_and_decode(monkeypatch):
    rng = np.random.default_rng(9)
    for (n, m, maxdeg) in [(1, 0, 1), (7, 3, 4), (65, 40, 6), (130, 90, 12), (300, 200, 5)]:
        rows = random_sparse_rows(rng, n, m, maxdeg)
        code = LinearCode.from_parity_check_matrix(n, rows)
        shots = 150
        words = np.stack([orc.sample_bsc_words(5, i, 0.08, n) for i in range(shots)]).reshape(shots, max(nwords(n), 1))

        def run():
            dec = BpDecoder(code, 0.08, 12)
            do = dev_u64(words.shape)
            dec.decode_batch(shots, to_dev_u64(words), do)
            return (host_u64(do),)

        pers, strm = _both(monkeypatch, run)
        assert _same(pers, strm), (n, m)


def test_stream_mc_counters_and_flags(monkeypatch):
    code = LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)

    def run():
        dec = BpDecoder(code, 0.06, 25)
        flags = dev_u8((5000,))
        cnt = dec.mc_run(0.06, 2024, 100, 5000, flags)
        return cnt, flags.cpu().numpy()

    pers, strm = _both(monkeypatch, run)
    assert _same(pers, strm)
    assert strm[0][CNT_SHOTS] == 5000 and int(strm[1].sum()) == strm[0][CNT_FAILURES]
    ref = oracle_code(code).mc_bp(0.06, 25, 2024, 100, 5000, nthreads=8)
    assert abs(int(strm[0][CNT_FAILURES]) - int(ref[CNT_FAILURES])) <= 2
    assert abs(int(strm[0][CNT_WORK]) - int(ref[CNT_WORK])) <= 20


def test_stream_split_invariance_and_empty(monkeypatch):
    monkeypatch.setenv("LDPC_HIP_BP_PATH", "stream")
    code = LinearCode.random_regular_code(96, 72, 3, 4, seed=2)
    dec = BpDecoder(code, 0.05, 30)
    total = dec.mc_run(0.05, 7, 0, 9000)
    a = dec.mc_run(0.05, 7, 0, 4321)
    b = dec.mc_run(0.05, 7, 4321, 9000 - 4321)
    assert ((a + b) == total).all()
    assert dec.mc_run(0.05, 7, 0, 0)[CNT_SHOTS] == 0
    assert dec.mc_run(0.05, 7, 0, 1)[CNT_SHOTS] == 1


def test_tail_handoff_changes_nothing(monkeypatch):
    """Persistent kernel with and without the tail hand-off (shots still running 16 iterations after the queue
    has drained are finished by the streamed kernels): identical outputs, iteration counts and counters, on a
    point with many non-converging shots."""
    code = LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)
    p, shots = 0.14, 3000
    oc = oracle_code(code)
    words = np.stack([orc.sample_bsc_words(77, i, p, 1200) for i in range(shots)])

    def run():
        dec = BpDecoder(code, p, 80)
        do, di = dev_u64(words.shape), dev_u32((shots,))
        dec.decode_batch(shots, to_dev_u64(words), do, di)
        flags = dev_u8((shots,))
        cnt = dec.mc_run(p, 77, 0, shots, flags)
        return host_u64(do), host_u32(di), cnt, flags.cpu().numpy()

    out = {}
    monkeypatch.setenv("LDPC_HIP_BP_PATH", "persistent")   # (a call this small would take the streamed form)
    for h in ("0", "16", "1"):
        monkeypatch.setenv("LDPC_HIP_BP_HANDOFF", h)
        out[h] = run()
    monkeypatch.delenv("LDPC_HIP_BP_HANDOFF")
    monkeypatch.delenv("LDPC_HIP_BP_PATH")
    assert (out["0"][1] == 80).sum() > 10          # the case has stragglers at the cap
    assert _same(out["0"], out["16"]) and _same(out["0"], out["1"])


def test_tail_compaction_changes_nothing(monkeypatch, capfd):
    """Once nothing is left to refill, the streamed form packs the shots still running into fewer groups
    (bps_compact: message columns, words, bit-sliced rows and slot state move to free slots).  Same corrections,
    iteration counts, posteriors, failure flags and counters as without it, on a point where most shots straggle."""
    code = LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)
    p, shots = 0.15, 2500
    oc = oracle_code(code)
    syn = np.stack([oc.syndrome_words(orc.sample_bsc_words(78, i, p, 1200)) for i in range(shots)])

    def run():
        dec = BpDecoder(code, p, 60)
        dc_, di, dp = dev_u64((shots, nwords(1200))), dev_u32((shots,)), dev_f64((shots, 1200))
        dec.correction_for_batch(shots, to_dev_u64(syn), dc_, di, dp)
        flags = dev_u8((shots,))
        cnt = dec.mc_run(p, 78, 0, shots, flags)
        return host_u64(dc_), host_u32(di), dp.cpu().numpy(), cnt, flags.cpu().numpy()

    monkeypatch.setenv("LDPC_HIP_DEBUG", "1")
    monkeypatch.setenv("LDPC_HIP_BP_PATH", "stream")
    capfd.readouterr()
    packed = run()
    assert "tail compactions" in capfd.readouterr().err
    monkeypatch.setenv("LDPC_HIP_BP_NO_COMPACT", "1")
    plain = run()
    assert "tail compactions" not in capfd.readouterr().err
    monkeypatch.delenv("LDPC_HIP_BP_NO_COMPACT")
    monkeypatch.setenv("LDPC_HIP_BP_PATH", "persistent")
    monkeypatch.setenv("LDPC_HIP_BP_HANDOFF", "4")
    handed = run()
    assert "tail compactions" in capfd.readouterr().err
    assert (plain[1] == 60).sum() > 20 and plain[4].sum() > 20        # stragglers at the cap, failures flagged
    assert _same(plain, packed) and _same(plain, handed)


def test_css_counters_same_on_every_path(monkeypatch):
    """CSS Monte-Carlo (X and Z halves, logical failure test): persistent kernel without hand-off, with hand-off,
    and the streamed form give the same counters."""
    from ldpc_amd import CssCode, CssDecoder
    base = LinearCode.random_regular_code(20, 15, 3, 4, seed=1)
    css = CssCode.hypergraph_product(base, base)
    res = []
    for env in ({"LDPC_HIP_BP_HANDOFF": "0"}, {"LDPC_HIP_BP_HANDOFF": "4"}, {"LDPC_HIP_BP_PATH": "stream"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        res.append(CssDecoder(css, 0.04, 60).mc_run(0.04, 5, 0, 6000))
        for k in env:
            monkeypatch.delenv(k)
    assert res[0][CNT_SHOTS] == 6000 and res[0][CNT_FAILURES] > 100
    assert (res[0] == res[1]).all() and (res[0] == res[2]).all()


def test_c5_full_size_properties():
    """BASELINE.json configs[4]: (3,6)-regular n = 10^6 (3 GB of messages per 64-shot group: the streamed form is the
    default there).  No oracle at this size in test time; size-independent properties instead: BP stops before the
    cap on every shot at p = 0.02, the correction reproduces the syndrome (H c = s), and since the code corrects
    this weight the correction IS the error (zero residual, no failures counted)."""
    from ldpc_amd import BinarySymmetricChannel
    n, m, shots, p = 1000000, 500000, 192, 0.02
    code = LinearCode.random_regular_code(n, m, 3, 6, seed=1)
    dec = BpDecoder(code, p, 100)
    de, ds = dev_u64((shots, nwords(n))), dev_u64((shots, nwords(m)))
    BinarySymmetricChannel(p).sample_batch(n, 31, 0, shots, de)
    code.syndrome_batch(shots, de, ds)
    dc, di = dev_u64((shots, nwords(n))), dev_u32((shots,))
    dec.correction_for_batch(shots, ds, dc, di)
    assert int(di.max()) < 100 and int(di.min()) >= 1
    chk = dev_u64((shots, nwords(m)))
    code.syndrome_batch(shots, dc, chk)
    assert bool((chk == ds).all())
    assert bool((dc == de).all())
    cnt = dec.mc_run(p, 31, 0, shots)
    assert cnt[CNT_SHOTS] == shots and cnt[CNT_FAILURES] == 0 and cnt[CNT_WORK] == int(di.sum())
