"""Flip decoder on the MI355X against the CPU oracle: bit-exact words, syndromes, flip counts."""
import numpy as np
import pytest

from ldpc_amd import (CNT_DETECTED, CNT_FAILURES, CNT_SHOTS, CNT_UNDETECTED, CNT_WORK, BinarySymmetricChannel,
                      FlipDecoder, LinearCode, nwords)
from oracle import oracle as orc
from tests.gpu_util import dev_u32, dev_u64, host_u32, host_u64, oracle_code, random_sparse_rows, to_dev_u64

pytestmark = pytest.mark.gpu

SEED = 2024


def test_flip_golden_c1():
    g = np.load("tests/golden/c1_vectors.npz")
    code = LinearCode.from_csr(1200, g["h_ptr"], g["h_idx"])
    dec = FlipDecoder(code)
    shots = g["errors"].shape[0]
    de, ds, do = dev_u64(g["errors"].shape), dev_u64(g["syndromes"].shape), dev_u64(g["flip_out"].shape)
    cnt = dec.mc_run(0.02, SEED, 0, shots, de, ds, do)
    assert (host_u64(de) == g["errors"]).all()
    assert (host_u64(ds) == g["syndromes"]).all()
    assert (host_u64(do) == g["flip_out"]).all()
    assert cnt[CNT_WORK] == g["flip_count"].sum() and cnt[CNT_SHOTS] == shots
    # the same shots through the batch entry point
    df, do2, ds2 = dev_u32((shots,)), dev_u64(g["flip_out"].shape), dev_u64(g["syndromes"].shape)
    dec.decode_batch(shots, to_dev_u64(g["errors"]), do2, ds2, df)
    assert (host_u64(do2) == g["flip_out"]).all() and (host_u64(ds2) == g["syndromes"]).all()
    assert (host_u32(df) == g["flip_count"]).all()


def test_flip_mc_counters_c1_full():
    # BASELINE configs[0]: (3,4) n=1200, p=0.02, 1e4 shots
    g = np.load("tests/golden/c1_vectors.npz")
    code = LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)
    assert (code.csr()[1] == g["h_idx"]).all()
    cnt = FlipDecoder(code).mc_run(0.02, SEED, 0, 10000)
    assert (cnt == g["c1_flip_counters_10000"]).all()


@pytest.mark.parametrize("n,m,dv,dc,p,shots", [(1200, 900, 3, 4, 0.02, 512), (10000, 5000, 3, 6, 0.02, 256),
                                               (10000, 5000, 3, 6, 0.05, 64), (4, 3, 3, 4, 0.2, 64),
                                               # more summary words than lanes per shot: the scan-cursor path
                                               (40000, 20000, 3, 6, 0.012, 24), (20000, 15000, 3, 4, 0.03, 24),
                                               # 21 and 13 neighbours: two rounds per flip / 7-check records
                                               (3000, 2000, 4, 6, 0.03, 96), (3000, 1500, 2, 4, 0.04, 96),
                                               # weight far above the error list: piecewise set-up
                                               (10000, 5000, 3, 6, 0.2, 24),
                                               # largest sizes: near the LDS limit of the fast path (4 shots per wave), and
                                               # the generic kernel near its own (one shot's state per wave, 125 of 160 KiB)
                                               (60000, 30000, 3, 6, 0.01, 8), (400000, 200000, 3, 6, 0.005, 4),
                                               # BASELINE configs[4]: the state of a shot (312 KB) exceeds the LDS of a CU and
                                               # lives in HBM (flip_big_kernel)
                                               (1000000, 500000, 3, 6, 0.005, 4), (1000000, 500000, 3, 6, 0.02, 2)])
def test_flip_matches_oracle(n, m, dv, dc, p, shots):
    code = LinearCode.random_regular_code(n, m, dv, dc, seed=1)
    oc = oracle_code(code)
    dec = FlipDecoder(code)
    de, ds, do = dev_u64((shots, nwords(n))), dev_u64((shots, nwords(m))), dev_u64((shots, nwords(n)))
    first = 777
    cnt = dec.mc_run(p, SEED, first, shots, de, ds, do)
    ge, gs, go = host_u64(de), host_u64(ds), host_u64(do)
    fails = det = flips = 0
    for i in range(shots):
        e = orc.sample_bsc_words(SEED, first + i, p, n)
        o, s_final, nf = oc.flip_decode_words(e)
        assert (ge[i] == e).all()
        assert (gs[i] == oc.syndrome_words(e)).all()
        assert (go[i] == o).all()
        fails += bool(o.any()); det += bool(s_final.any()); flips += nf
    assert cnt[CNT_SHOTS] == shots and cnt[CNT_FAILURES] == fails and cnt[CNT_DETECTED] == det
    assert cnt[CNT_UNDETECTED] == fails - det and cnt[CNT_WORK] == flips


@pytest.mark.parametrize("n,m,dv,dc,p,shots", [(10000, 5000, 3, 6, 0.02, 70001), (1200, 900, 3, 4, 0.02, 200000),
                                               (10000, 5000, 3, 6, 0.0004, 66000), (15000, 7500, 3, 6, 0.015, 66000)])
def test_flip_large_runs_take_the_service_wave_form_and_change_nothing(monkeypatch, n, m, dv, dc, p, shots):
    """Runs of 65536 shots or more on regular codes of bit degree 3 go through flip4_kernel (flip waves fed by
    service waves); smaller ones, and LDPC_HIP_FLIP_KERNEL=v3, through flip3_kernel.  Every recorded word must be the
    same, with and without the caller's error buffer, and equal to the oracle's (flip.rs:21-43) on a sample;
    counters equal to the oracle's over the whole run."""
    code = LinearCode.random_regular_code(n, m, dv, dc, seed=1)
    oc = oracle_code(code)
    dec = FlipDecoder(code)
    first = 12345
    de, ds, do = dev_u64((shots, nwords(n))), dev_u64((shots, nwords(m))), dev_u64((shots, nwords(n)))
    cnt = dec.mc_run(p, SEED, first, shots, de, ds, do)
    do_b = dev_u64((shots, nwords(n)))
    cnt_b = dec.mc_run(p, SEED, first, shots, None, None, do_b)     # parked words in the library's scratch
    assert (cnt_b == cnt).all() and bool((do_b == do).all())
    monkeypatch.setenv("LDPC_HIP_FLIP_KERNEL", "v3")
    de3, ds3, do3 = dev_u64((shots, nwords(n))), dev_u64((shots, nwords(m))), dev_u64((shots, nwords(n)))
    cnt3 = dec.mc_run(p, SEED, first, shots, de3, ds3, do3)
    monkeypatch.delenv("LDPC_HIP_FLIP_KERNEL")
    assert (cnt3 == cnt).all()
    assert bool((de3 == de).all()) and bool((ds3 == ds).all()) and bool((do3 == do).all())
    ref = oc.mc_flip(p, SEED, first, shots, nthreads=16)
    assert (cnt == ref).all(), (cnt, ref)
    ge, gs, go = host_u64(de), host_u64(ds), host_u64(do)
    for i in list(range(0, shots, max(shots // 150, 1))) + [shots - 1]:
        e = orc.sample_bsc_words(SEED, first + i, p, n)
        o, _, _ = oc.flip_decode_words(e)
        assert (ge[i] == e).all() and (gs[i] == oc.syndrome_words(e)).all() and (go[i] == o).all()


def test_flip_decode_batch_large_goes_through_the_service_wave_form():
    """decode_batch (caller's words in, flip counts out) on 70,000 words takes flip4_kernel like a large Monte-Carlo
    run: decoding the errors a run recorded reproduces its decoded words, syndromes and the flip total."""
    n, m, shots, p = 10000, 5000, 70000, 0.02
    code = LinearCode.random_regular_code(n, m, 3, 6, seed=1)
    dec = FlipDecoder(code)
    de, ds, do = dev_u64((shots, nwords(n))), dev_u64((shots, nwords(m))), dev_u64((shots, nwords(n)))
    cnt = dec.mc_run(p, SEED, 5, shots, de, ds, do)
    do2, ds2, df = dev_u64((shots, nwords(n))), dev_u64((shots, nwords(m))), dev_u32((shots,))
    dec.decode_batch(shots, de, do2, ds2, df)
    assert bool((do2 == do).all()) and bool((ds2 == ds).all())
    assert int(df.sum()) == cnt[CNT_WORK]
    oc = oracle_code(code)
    gf, ge = host_u32(df), host_u64(de)
    for i in (0, 1, 34567, shots - 1):
        assert gf[i] == oc.flip_decode_words(ge[i])[2]


def test_flip_irregular_and_ragged():
    rng = np.random.default_rng(3)
    shapes = [(1, 0), (1, 1), (63, 10), (64, 64), (65, 33), (129, 200), (300, 7)]
    for (n, m) in shapes:
        rows = random_sparse_rows(rng, n, m, 8)
        code = LinearCode.from_parity_check_matrix(n, rows)
        oc = oracle_code(code)
        dec = FlipDecoder(code)
        shots = 48
        words = np.zeros((shots, nwords(n)), dtype=np.uint64)
        for i in range(shots):
            words[i] = orc.sample_bsc_words(9, i, 0.2, n)
        do, ds, df = dev_u64((shots, nwords(n))), dev_u64((shots, max(nwords(m), 1))), dev_u32((shots,))
        dec.decode_batch(shots, to_dev_u64(words), do, ds, df)
        go, gf = host_u64(do), host_u32(df)
        for i in range(shots):
            o, _, nf = oc.flip_decode_words(words[i])
            assert (go[i] == o).all() and gf[i] == nf
            # and the reference-style restatement on position lists
            ref, nf2 = oc.flip_decode(orc.words_to_pos(words[i], n))
            assert orc.words_to_pos(go[i], n) == ref and nf2 == nf


def test_flip_full_size_properties():
    # C2 shape, 2^16 shots: decode is idempotent, the recorded syndrome is H*error,
    # counters add up over a split shot range (noise is keyed by the global shot index)
    n, m, shots = 10000, 5000, 1 << 16
    code = LinearCode.random_regular_code(n, m, 3, 6, seed=1)
    dec = FlipDecoder(code)
    de, ds, do = dev_u64((shots, nwords(n))), dev_u64((shots, nwords(m))), dev_u64((shots, nwords(n)))
    cnt = dec.mc_run(0.02, SEED, 0, shots, de, ds, do)
    chk = dev_u64((shots, nwords(m)))
    code.syndrome_batch(shots, de, chk)
    assert bool((chk == ds).all())
    do2, df = dev_u64((shots, nwords(n))), dev_u32((shots,))
    dec.decode_batch(shots, do, do2, None, df)
    assert bool((do2 == do).all()) and int(df.sum()) == 0
    # failures counted == decoded words that are non-zero
    assert int((do != 0).any(dim=1).sum()) == cnt[CNT_FAILURES]
    code.syndrome_batch(shots, do, chk)
    assert int((chk != 0).any(dim=1).sum()) == cnt[CNT_DETECTED]
    a = dec.mc_run(0.02, SEED, 0, shots // 2)
    b = dec.mc_run(0.02, SEED, shots // 2, shots - shots // 2)
    assert ((a + b) == cnt).all()
    mean_w = float(sum(bin(int(x)).count("1") for x in host_u64(de[:64]).ravel())) / 64
    assert 150 < mean_w < 250


def test_single_process_multi_device_runs_sum_to_the_single_run():
    """ldpc_*_mc_run_multi: one host thread per device, shots sharded by global index, counters summed on the host.
    On a one-GPU box the device list names GPU 0 several times (the shards then run concurrently on it)."""
    from ldpc_amd import BpDecoder, CssCode, CssDecoder, mc_run_multi
    code = LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)
    ref = FlipDecoder(code).mc_run(0.04, 5, 100, 30001)
    for devs in ([0], [0, 0], [0, 0, 0]):
        assert (mc_run_multi(code, devs, 0.04, 5, 100, 30001, "flip") == ref).all()
    ref = BpDecoder(code, 0.09, 20).mc_run(0.09, 5, 100, 5000)
    assert (mc_run_multi(code, [0, 0], 0.09, 5, 100, 5000, "bp", 20) == ref).all()
    css = CssCode.shor_code()
    ref = CssDecoder(css, 0.05, 30).mc_run(0.05, 5, 0, 4000)
    assert (mc_run_multi(css, [0, 0], 0.05, 5, 0, 4000, "css", 30) == ref).all()
    with pytest.raises(Exception):
        mc_run_multi(code, [99], 0.04, 5, 0, 10, "flip")


def test_flip_beyond_the_lds_limit_runs_from_hbm_and_only_the_no_alloc_call_refuses_it():
    """n = 10^6 (BASELINE configs[4]): a shot's state does not fit a CU's 160 KiB of LDS; FlipDecoder::decode has no
    size limit (flip.rs:21-33), so the library keeps that state in HBM (flip_big_kernel; parity in
    test_flip_matches_oracle).  What truly cannot be served is the same code through ldpc_flip_mc_run_noalloc, whose
    static scratch array is not sized for it: LDPC_ERR_CAPACITY, no launch."""
    import torch
    import ldpc_amd
    code = LinearCode.random_regular_code(1000000, 500000, 3, 6, seed=1)
    dec = FlipDecoder(code)
    cnt = dec.mc_run(0.001, SEED, 0, 8)
    assert cnt[CNT_SHOTS] == 8
    # decode_batch on the decoded words: idempotent (nothing left to flip)
    do = dev_u64((8, nwords(1000000)))
    dec.mc_run(0.001, SEED, 0, 8, None, None, do)
    do2, df = dev_u64((8, nwords(1000000))), dev_u32((8,))
    dec.decode_batch(8, do, do2, None, df)
    assert bool((do2 == do).all()) and int(df.sum()) == 0
    with pytest.raises(ldpc_amd.LdpcError) as e:
        dec.mc_run_noalloc(0.001, SEED, 0, 4, torch.zeros(8, dtype=torch.int64, device="cuda"))
    assert e.value.status == ldpc_amd.ERR_CAPACITY
