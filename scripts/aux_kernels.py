"""Throughput of the batched noise / syndrome kernels at C2; diagnostics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldpc_amd
from ldpc_amd import BinarySymmetricChannel, DepolarizingNoise, LinearCode, nwords

n, m, shots = 10000, 5000, 1 << 20
code = LinearCode.random_regular_code(n, m, 3, 6, seed=1)
wn, wm = nwords(n), nwords(m)
de = torch.empty((shots, wn), dtype=torch.int64, device="cuda")
dz = torch.empty((shots, wn), dtype=torch.int64, device="cuda")
ds = torch.empty((shots, wm), dtype=torch.int64, device="cuda")


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for p in (0.02, 0.001):
    t = timed(lambda: BinarySymmetricChannel(p).sample_batch(n, 7, 0, shots, de))
    print("sample_bsc p=%g: %.2f ms, %.3g shots/s, %.0f GB/s written" % (p, t * 1e3, shots / t, shots * wn * 8 / t / 1e9))
t = timed(lambda: DepolarizingNoise(0.01).sample_batch(n, 7, 0, shots, de, dz))
print("sample_depolarizing p=0.01: %.2f ms, %.3g shots/s, %.0f GB/s written" % (t * 1e3, shots / t, 2 * shots * wn * 8 / t / 1e9))
BinarySymmetricChannel(0.02).sample_batch(n, 7, 0, shots, de)
t = timed(lambda: code.syndrome_batch(shots, de, ds))
print("syndrome H*e: %.2f ms, %.3g shots/s, %.0f GB/s (words read + syndromes written)" % (t * 1e3, shots / t, shots * (wn + wm) * 8 / t / 1e9))
