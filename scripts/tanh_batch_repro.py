"""Diagnostic (round 3): the tanh batch for bits of any degree up to eight, dropped in round 2 for a wrong result in one
irregular case.  Runs the LDS form of a variant library (AB_LIB, built with -DBPL_TANH_BATCH_ANY=1|2) against the
persistent kernel on the irregular codes of tests/test_gpu_bp_lds.py and reports where they part: shot, bits, the
degrees of those bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ldpc_amd
if os.environ.get("AB_LIB"):
    ldpc_amd.LIB_PATH = os.environ["AB_LIB"]
from ldpc_amd import BpDecoder, LinearCode, nwords
from oracle import oracle as orc
from tests.gpu_util import dev_u64, dev_u32, host_u64, host_u32, random_sparse_rows, to_dev_u64


def codes():
    rng = np.random.default_rng(17)
    n, m = 150, 120
    hot = rng.choice(n - 10, size=6, replace=False)
    rows = []
    for c in range(m):
        pool = np.concatenate([hot, rng.choice(n - 10, size=8, replace=False)]) if c % 2 == 0 else rng.choice(n - 10, size=12, replace=False)
        rows.append(sorted(int(x) for x in rng.choice(np.unique(pool), size=4, replace=False)))
    yield "regular checks, bits 0..>8", LinearCode.from_parity_check_matrix(n, rows), rows, n
    rng = np.random.default_rng(9)
    for (n, m, maxdeg) in [(7, 3, 4), (65, 40, 6), (130, 90, 12), (300, 200, 5)]:
        rows = random_sparse_rows(rng, n, m, maxdeg)
        yield "irregular rows n=%d" % n, LinearCode.from_parity_check_matrix(n, rows), rows, n
    rng = np.random.default_rng(5)
    for (n, m, maxdeg) in [(40, 30, 3), (200, 150, 9)]:
        rows = random_sparse_rows(rng, n, m, maxdeg)
        yield "irregular rows n=%d" % n, LinearCode.from_parity_check_matrix(n, rows), rows, n


def run(code, n, words, path, cap):
    os.environ["LDPC_HIP_BP_PATH"] = path
    ldpc_amd.reload_switches()
    do, di = dev_u64(words.shape), dev_u32((words.shape[0],))
    BpDecoder(code, 0.08, cap).decode_batch(words.shape[0], to_dev_u64(words), do, di)
    name = ldpc_amd.last_kernel_name()
    return host_u64(do), host_u32(di), name


bad = 0
for label, code, rows, n in codes():
    degs = np.bincount(np.concatenate([np.array(r, dtype=np.int64) for r in rows if len(r)] or [np.zeros(0, np.int64)]), minlength=n)
    shots = 400
    words = np.stack([orc.sample_bsc_words(5, i, 0.08, n) for i in range(shots)]).reshape(shots, max(nwords(n), 1))
    for cap in (1, 2, 3, 12, 20):
        a, ia, ka = run(code, n, words, "lds", cap)
        b, ib, kb = run(code, n, words, "persistent", cap)
        diff = np.nonzero((a != b).any(axis=1) | (ia != ib))[0]
        if len(diff):
            bad += 1
            i = int(diff[0])
            bits = np.nonzero(np.unpackbits((a[i] ^ b[i]).view(np.uint8), bitorder="little")[:n])[0]
            print("%-28s cap %2d: %d of %d shots differ (%s vs %s); first shot %d, iterations %d vs %d, bits %s of degrees %s; degree histogram %s"
                  % (label, cap, len(diff), shots, ka, kb, i, ia[i], ib[i], bits[:8].tolist(), degs[bits[:8]].tolist(),
                     np.bincount(degs).tolist()), flush=True)
        else:
            print("%-28s cap %2d: equal (%s)" % (label, cap, ka), flush=True)
print("cases that differ:", bad)
