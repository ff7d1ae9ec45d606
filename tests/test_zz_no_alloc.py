"""ldpc_flip_mc_run_noalloc (the entry point that allocates nothing: scratch in a static device array) against
ldpc_flip_mc_run, through the C ABI: counters and every recorded word bit for bit, on two calls in a row, the second
with a smaller input, then on other codes (other kernels share the array).  Collected last (test_zz_*): a failure
here must not hide other tests."""
import numpy as np
import pytest
import torch

import ldpc_amd
from ldpc_amd import FlipDecoder, LinearCode, nwords
from tests.gpu_util import dev_u64, host_u64

pytestmark = pytest.mark.gpu

SEED = 2024


def _both(dec, n, m, p, first, shots, record=True):
    de, ds, do = dev_u64((shots, nwords(n))), dev_u64((shots, max(nwords(m), 1))), dev_u64((shots, nwords(n)))
    ref = dec.mc_run(p, SEED, first, shots, de if record else None, ds if record else None, do)
    de2, ds2, do2 = dev_u64((shots, nwords(n))), dev_u64((shots, max(nwords(m), 1))), dev_u64((shots, nwords(n)))
    dcnt = torch.full((8,), -1, dtype=torch.int64, device="cuda")
    dec.mc_run_noalloc(p, SEED, first, shots, dcnt, de2 if record else None, ds2 if record else None, do2)
    torch.cuda.synchronize()
    got = dcnt.cpu().numpy().view(np.uint64)
    assert (got == ref).all(), (got, ref)
    assert bool((do2 == do).all())
    if record:
        assert bool((de2 == de).all()) and bool((ds2 == ds).all())
    return ref


def test_no_alloc_entry_point_matches_the_allocating_one():
    n, m = 10000, 5000
    code = LinearCode.random_regular_code(n, m, 3, 6, seed=1)
    dec = FlipDecoder(code)
    with pytest.raises(ldpc_amd.LdpcError):      # not set up on a device yet: refused, no allocation behind the caller's back
        dec.mc_run_noalloc(0.02, SEED, 0, 8, torch.zeros(8, dtype=torch.int64, device="cuda"))
    dec.mc_run(0.02, SEED, 0, 64)                # first ordinary call uploads graph and tables
    a = _both(dec, n, m, 0.02, 1000, 20000)
    b = _both(dec, n, m, 0.02, 5, 3000)          # second call, smaller: the array still holds the first call's logs
    assert a[0] == 20000 and b[0] == 3000
    _both(dec, n, m, 0.02, 77, 4099, record=False)   # parked words in the array instead of the caller's buffer
    _both(dec, n, m, 0.0, 0, 100)
    # other codes, other kernels (flip2_kernel: bit degree 4; flip_kernel: degree 9 bits), same array
    for (nn, mm, dv, dc, p, shots) in [(3000, 2000, 4, 6, 0.03, 500), (1200, 900, 3, 4, 0.02, 2048), (300, 900, 9, 3, 0.01, 64)]:
        c2 = LinearCode.random_regular_code(nn, mm, dv, dc, seed=3)
        d2 = FlipDecoder(c2)
        d2.mc_run(p, SEED, 0, 8)
        _both(d2, nn, mm, p, 10, shots)
        _both(d2, nn, mm, p, 0, shots // 3)
    # a code on the generic kernel (n = 400,000: a shot's state fills most of a CU's LDS, no scratch pieces at all):
    # the entry point serves it, bit for bit like the allocating call.  (The refusal of a code beyond the array,
    # n = 10^6, is tested in tests/test_gpu_flip.py.)
    big = LinearCode.random_regular_code(400000, 200000, 3, 6, seed=1)
    dbig = FlipDecoder(big)
    dbig.mc_run(0.001, SEED, 0, 4)
    _both(dbig, 400000, 200000, 0.001, 0, 4)
