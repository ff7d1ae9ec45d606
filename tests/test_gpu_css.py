"""CSS path (examples/css_decoding.rs) on the MI355X against the CPU oracle."""
import numpy as np
import pytest

from ldpc_amd import CNT_FAILURES, CNT_SHOTS, CNT_WORK, CNT_XFAIL, CNT_ZFAIL, CssCode, CssDecoder, LinearCode
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

SEED = 2024


def _example_code():
    base = LinearCode.random_regular_code(20, 15, 3, 4, seed=1)     # css_decoding.rs:25-34
    return CssCode.hypergraph_product(base, base)


def _close(cnt, ref, shots):
    # float path: allow 1e-4 of the shots (north_star) to decide differently
    tol = max(2, int(round(1e-4 * shots)))
    assert cnt[CNT_SHOTS] == shots
    for k in (CNT_FAILURES, CNT_XFAIL, CNT_ZFAIL):
        assert abs(int(cnt[k]) - int(ref[k])) <= tol, (k, cnt, ref)
    assert abs(int(cnt[CNT_WORK]) - int(ref[CNT_WORK])) <= 200 * tol


def test_css_mc_matches_golden():
    g = np.load("tests/golden/c1_vectors.npz")
    css = _example_code()
    assert (css.x_stabs_binary().csr()[1] == g["css_hx_idx"]).all()
    assert (css.x_logicals_binary().csr()[1] == g["css_lx_idx"]).all()
    for p in (0.01, 0.05):
        dec = CssDecoder(css, p, 100)
        cnt = dec.mc_run(p, SEED, 0, 2000)
        _close(cnt, g["css_counters_p%g" % p], 2000)


def test_css_single_shot_matches_oracle():
    css = _example_code()
    n = css.len()
    mats = [css.x_stabs_binary(), css.z_stabs_binary(), css.x_logicals_binary(), css.z_logicals_binary()]
    hx, hz, lx, lz = [orc.Code(n, ptr=m.csr()[0].astype(np.int64), idx=m.csr()[1].astype(np.uint32)) for m in mats]
    dec = CssDecoder(css, 0.03, 100)
    agree = 0
    shots = 40
    for shot in range(shots):
        ex, ez = orc.sample_depolarizing_words(SEED, shot, 0.03, n)
        xp, zp = orc.words_to_pos(ex, n), orc.words_to_pos(ez, n)
        sx, sz = css.syndrome_of(xp, zp)
        osx, osz = orc.css_syndrome_words(hx, hz, ex, ez)
        assert sx == orc.words_to_pos(osx, hx.m) and sz == orc.words_to_pos(osz, hz.m)
        cx, cz = dec.correction_for(sx, sz)
        ocz, _, _ = hx.bp_correction_for(0.03, 100, osx)     # decoders/css.rs: z correction from the x decoder
        ocx, _, _ = hz.bp_correction_for(0.03, 100, osz)
        same = cx == orc.words_to_pos(ocx, n) and cz == orc.words_to_pos(ocz, n)
        agree += same
        if same:
            rx, rz = sorted(set(xp) ^ set(cx)), sorted(set(zp) ^ set(cz))
            fails = orc.css_residual_fails(hx, hz, lx, lz, orc.pos_to_words(rx, n), orc.pos_to_words(rz, n))
            assert css.has_stabilizer(rx, rz) == (fails == 0)
    assert agree >= shots - 1


def test_css_mc_split_invariance_and_shor():
    css = CssCode.shor_code()
    dec = CssDecoder(css, 0.05, 30)
    total = dec.mc_run(0.05, 3, 0, 5000)
    a = dec.mc_run(0.05, 3, 0, 1700)
    b = dec.mc_run(0.05, 3, 1700, 3300)
    assert ((a + b) == total).all()
    d = {"hx": css.x_stabs_binary(), "hz": css.z_stabs_binary(), "lx": css.x_logicals_binary(),
         "lz": css.z_logicals_binary()}
    oc = {k: orc.Code(9, rows=v.parity_check_matrix()) for k, v in d.items()}
    ref = orc.mc_css_bp(oc["hx"], oc["hz"], oc["lx"], oc["lz"], 0.05, 30, 3, 0, 5000, nthreads=8)
    _close(total, ref, 5000)


def test_css_halves_side_by_side_or_in_turn(monkeypatch):
    """Small runs decode the X and Z halves at once (two host threads, two streams, separate flag bytes); large runs
    one after the other.  Same counters either way, for a run of each kind and with the choice forced."""
    css = _example_code()
    for shots, p in ((3000, 0.03), (20000, 0.02)):
        res = []
        for side in (None, "0", "1"):
            if side is None:
                monkeypatch.delenv("LDPC_HIP_CSS_SIDE_BY_SIDE", raising=False)
            else:
                monkeypatch.setenv("LDPC_HIP_CSS_SIDE_BY_SIDE", side)
            res.append(CssDecoder(css, p, 40).mc_run(p, 9, 17, shots))
        monkeypatch.delenv("LDPC_HIP_CSS_SIDE_BY_SIDE", raising=False)
        assert res[0][CNT_SHOTS] == shots and res[0][CNT_FAILURES] > 20
        assert res[0][CNT_XFAIL] > 0 and res[0][CNT_ZFAIL] > 0
        assert (res[0] == res[1]).all() and (res[0] == res[2]).all()
