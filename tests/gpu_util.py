"""Helpers for the GPU parity tests: device buffers come from torch (plumbing),
the hot path is always called through the C ABI via ldpc_amd."""
import numpy as np


def dev_u64(shape):
    import torch
    return torch.zeros(shape, dtype=torch.int64, device="cuda")


def dev_u32(shape):
    import torch
    return torch.zeros(shape, dtype=torch.int32, device="cuda")


def dev_u8(shape):
    import torch
    return torch.zeros(shape, dtype=torch.uint8, device="cuda")


def dev_f64(shape):
    import torch
    return torch.zeros(shape, dtype=torch.float64, device="cuda")


def to_dev_u64(arr):
    import torch
    a = np.ascontiguousarray(arr, dtype=np.uint64).view(np.int64)
    return torch.from_numpy(a.copy()).to("cuda")


def host_u64(t):
    return t.cpu().numpy().view(np.uint64)


def host_u32(t):
    return t.cpu().numpy().view(np.uint32)


def oracle_code(code):
    """Oracle handle for the same parity-check matrix as an ldpc_amd.LinearCode."""
    from oracle import oracle as orc
    ptr, idx = code.csr()
    return orc.Code(code.len(), ptr=ptr.astype(np.int64), idx=idx.astype(np.uint32))


def random_sparse_rows(rng, n, m, max_deg):
    rows = []
    for _ in range(m):
        k = int(rng.integers(0, min(n, max_deg) + 1))
        rows.append(sorted(int(x) for x in rng.choice(n, size=k, replace=False)))
    return rows


def agree_f64(a, b, rtol=1e-5):
    """rows of a, b agree: same non-finite pattern (NaN = NaN, inf = inf with sign), finite values within rtol"""
    fa, fb = np.isfinite(a), np.isfinite(b)
    same_kind = (fa == fb) & (np.isnan(a) == np.isnan(b)) & (fa | np.isnan(a) | (a == b))
    close = np.isclose(np.where(fa, a, 0.0), np.where(fb, b, 0.0), rtol=rtol, atol=0.0)
    return (same_kind & (close | ~fa)).all(axis=1)


def assert_bp_matches_oracle(code, oc, p, cap, syn, gc, gi, nthreads=16, need=0.9999, follow=4):
    """North-star bar for a float decoder, asserted (never skipped): corrections gc / iteration counts gi of a GPU run
    on the syndromes `syn` against the oracle's restatement of belief_propagation.rs:23-30 on the same syndromes.
    * decisions equal on >= `need` of the shots the oracle converges on (and on >= 99 % of all shots: a run that the cap
      stops carries NaN / inf messages whose rounding noise may tip a bit);
    * iteration counts equal wherever decisions are;
    * every differing shot (up to `follow`) is followed: its check -> bit messages after iteration 1, 2, 3 must agree
      with the oracle's at rtol 1e-5 (ldpc_bp_correction_for_batch_messages), i.e. the split is late rounding noise,
      not arithmetic.
    Returns the share of equal shots."""
    from ldpc_amd import BpDecoder, nwords
    rc, ri, _, _ = oc.bp_correction_for_batch(p, cap, syn, nthreads=nthreads)
    same = (gc == rc).all(axis=1)
    conv = ri < cap
    if conv.any():
        assert same[conv].mean() >= need, ("converging shots that differ", int((~same[conv]).sum()), int(conv.sum()))
    assert same.mean() >= 0.99, int((~same).sum())
    assert (gi[same] == ri[same]).all()
    n, E = code.len(), code.num_edges()
    for i in np.nonzero(~same)[0][:follow]:
        for k in (1, 2, 3):
            if k > cap:
                break
            dec = BpDecoder(code, p, k)
            dc, di = dev_u64((1, max(nwords(n), 1))), dev_u32((1,))
            dp, dm = dev_f64((1, max(n, 1))), dev_f64((1, max(E, 1)))
            dec.correction_for_batch(1, to_dev_u64(syn[i : i + 1]), dc, di, dp, dm)
            _, rit, _, rm = oc.bp_correction_for_batch(p, k, syn[i : i + 1], want_messages=True)
            assert host_u32(di)[0] == rit[0] and agree_f64(dm.cpu().numpy(), rm).all(), (int(i), k)
    return float(same.mean())
