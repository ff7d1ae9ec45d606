"""Device noise sampling and batched syndromes against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from ldpc_amd import BinarySymmetricChannel, DepolarizingNoise, LinearCode, nwords
from oracle import oracle as orc
from tests.gpu_util import dev_u64, host_u64, oracle_code, random_sparse_rows, to_dev_u64

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,p", [(1200, 0.02), (10000, 0.02), (1, 0.5), (63, 0.3), (64, 0.3), (65, 0.3),
                                 (130, 0.0), (130, 1.0), (700, 0.9), (4, 0.2)])
def test_bsc_batch_matches_oracle(n, p):
    shots, first, seed = 96, 12345678901, 2024
    d = dev_u64((shots, nwords(n)))
    BinarySymmetricChannel(p).sample_batch(n, seed, first, shots, d)
    got = host_u64(d)
    for i in range(shots):
        assert (got[i] == orc.sample_bsc_words(seed, first + i, p, n)).all()


def test_bsc_single_sample_api():
    noise = BinarySymmetricChannel.with_probability(0.2)
    code = LinearCode.random_regular_code(4, 3, 3, 4, seed=1)      # examples/random_linear_code.rs
    err = code.random_error(noise, seed=7, shot=0)
    assert err == orc.words_to_pos(orc.sample_bsc_words(7, 0, 0.2, 4), 4)
    assert noise.sample_error_of_length(0, 7, 0) == []


@pytest.mark.parametrize("n,p", [(625, 0.01), (625, 0.3), (7, 1.0), (200, 0.0)])
def test_depolarizing_batch_matches_oracle(n, p):
    shots, seed = 64, 99
    dx, dz = dev_u64((shots, nwords(n))), dev_u64((shots, nwords(n)))
    DepolarizingNoise(p).sample_batch(n, seed, 0, shots, dx, dz)
    gx, gz = host_u64(dx), host_u64(dz)
    for i in range(shots):
        ox, oz = orc.sample_depolarizing_words(seed, i, p, n)
        assert (gx[i] == ox).all() and (gz[i] == oz).all()
    pos, paulis = DepolarizingNoise(p).sample_error_of_length(n, seed, 3)
    ox, oz = orc.sample_depolarizing_words(seed, 3, p, n)
    xs, zs = set(orc.words_to_pos(ox, n)), set(orc.words_to_pos(oz, n))
    assert pos == sorted(xs | zs)
    for q, pl in zip(pos, paulis):
        assert pl == ("Y" if q in xs and q in zs else "X" if q in xs else "Z")


def test_depolarizing_golden():
    g = np.load("tests/golden/c1_vectors.npz")
    dx, dz = dev_u64(g["depol_x"].shape), dev_u64(g["depol_z"].shape)
    DepolarizingNoise(0.05).sample_batch(625, 2024, 0, g["depol_x"].shape[0], dx, dz)
    assert (host_u64(dx) == g["depol_x"]).all() and (host_u64(dz) == g["depol_z"]).all()


def test_syndrome_batch_matches_oracle():
    rng = np.random.default_rng(11)
    cases = [LinearCode.random_regular_code(1200, 900, 3, 4, seed=1), LinearCode.hamming_code()]
    for _ in range(6):
        n, m = int(rng.integers(1, 300)), int(rng.integers(0, 200))
        cases.append(LinearCode.from_parity_check_matrix(n, random_sparse_rows(rng, n, m, 9)))
    for code in cases:
        n, m = code.len(), code.num_checks()
        oc = oracle_code(code)
        shots = 40
        words = np.zeros((shots, max(nwords(n), 1)), dtype=np.uint64)
        for i in range(shots):
            words[i, : nwords(n)] = orc.sample_bsc_words(5, i, 0.1, n)
        dw = to_dev_u64(words[:, : nwords(n)])
        ds = dev_u64((shots, max(nwords(m), 1)))
        code.syndrome_batch(shots, dw, ds)
        got = host_u64(ds)
        for i in range(shots):
            assert (got[i, : nwords(m)] == oc.syndrome_words(words[i, : nwords(n)])).all()


def test_syndrome_linearity_full_size():
    # size-independent property at the C2 shape: H(a ^ b) = Ha ^ Hb
    code = LinearCode.random_regular_code(10000, 5000, 3, 6, seed=1)
    shots = 4096
    wn, wm = nwords(10000), nwords(5000)
    a, b = dev_u64((shots, wn)), dev_u64((shots, wn))
    BinarySymmetricChannel(0.02).sample_batch(10000, 1, 0, shots, a)
    BinarySymmetricChannel(0.3).sample_batch(10000, 2, 0, shots, b)
    sa, sb, sab = dev_u64((shots, wm)), dev_u64((shots, wm)), dev_u64((shots, wm))
    code.syndrome_batch(shots, a, sa)
    code.syndrome_batch(shots, b, sb)
    code.syndrome_batch(shots, a ^ b, sab)
    assert bool(((sa ^ sb) == sab).all())
    assert int((sa != 0).sum()) > 0


def test_erasure_channel_is_the_bernoulli_stream():
    """ErasureChannel (noise/erasure.rs:33-42) draws the same n i.i.d. Bernoulli(p) bits as the BSC."""
    from ldpc_amd import BinarySymmetricChannel, ErasureChannel
    for p, n in ((0.1, 500), (0.0, 40), (1.0, 70)):
        for shot in (0, 5):
            e = ErasureChannel.with_probability(p).sample_error_of_length(n, 9, shot)
            assert e == BinarySymmetricChannel(p).sample_error_of_length(n, 9, shot)
            assert e == sorted(set(e)) and all(0 <= x < n for x in e)
    assert str(ErasureChannel(0.25)) == "Erasure(0.25)"


def test_small_probability_rate():
    """the gap table quantises to 32 bits; at the smallest probabilities it accepts the sampled rate must
    still be the channel's (mean weight over many shots within 4 sigma), and below 1e-7 the call is refused."""
    import torch
    n, shots = 10000, 1 << 17
    for p in (2e-5, 1e-6, 1.2e-7):
        d = dev_u64((shots, nwords(n)))
        BinarySymmetricChannel(p).sample_batch(n, 99, 0, shots, d)
        w = d.view(torch.uint8).cpu().numpy()
        weight = int(np.unpackbits(w).sum())
        mean = n * shots * p
        assert abs(weight - mean) <= 4 * mean ** 0.5 + 1, (p, weight, mean)
