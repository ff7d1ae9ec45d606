"""Parity stress of the LDS-resident BP form against the persistent kernel on assorted codes, channels and caps
(diagnostics): corrections, iteration counts and Monte-Carlo counters must be identical.  usage: bp_lds_stress.py [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd
import numpy as np
from ldpc_amd import BpDecoder, CssCode, CssDecoder, LinearCode, nwords
from oracle import oracle as orc
from tests.gpu_util import dev_u32, dev_u64, host_u32, host_u64, oracle_code, random_sparse_rows, to_dev_u64

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(100 + seed)
cases = []
for (n, m, dv, dc) in [(1200, 900, 3, 4), (700, 300, 3, 7), (600, 300, 3, 6), (300, 300, 3, 3), (2000, 1000, 3, 6), (800, 400, 4, 8),
                       (400, 200, 2, 4), (960, 240, 1, 4), (630, 270, 3, 7), (1500, 900, 3, 5), (64, 48, 3, 4), (65, 39, 3, 5)]:
    cases.append(("reg(%d,%d) n=%d" % (dv, dc, n), LinearCode.random_regular_code(n, m, dv, dc, seed=3 + seed)))
for (n, m, maxdeg) in [(500, 300, 9), (900, 700, 14), (129, 64, 4), (1000, 400, 3), (63, 63, 6)]:
    cases.append(("irregular n=%d m=%d deg<=%d" % (n, m, maxdeg), LinearCode.from_parity_check_matrix(n, random_sparse_rows(rng, n, m, maxdeg))))
cases.append(("repetition 33", LinearCode.repetition_code(33)))
cases.append(("hamming", LinearCode.hamming_code()))


def forms(fn):
    out = []
    for path in ("lds", "persistent"):
        os.environ["LDPC_HIP_BP_PATH"] = path
        ldpc_amd.reload_switches()
        out.append(fn())
        name = ldpc_amd.last_kernel_name()
        assert name.startswith("bp_lds_kernel") == (path == "lds"), (path, name)
    return out


bad = 0
for name, code in cases:
    n, m = code.len(), code.num_checks()
    oc = oracle_code(code)
    for p, cap in ((0.01, 100), (0.05, 31), (0.12, 14), (0.65, 9)):
        shots = 400
        words = np.stack([orc.sample_bsc_words(4 + seed, i, min(p, 0.1), n) for i in range(shots)]).reshape(shots, max(nwords(n), 1))
        syn = np.stack([oc.syndrome_words(words[i]) for i in range(shots)]).reshape(shots, max(nwords(m), 1))

        def run():
            dec = BpDecoder(code, p, cap)
            dc_, di = dev_u64(words.shape), dev_u32((shots,))
            dec.correction_for_batch(shots, to_dev_u64(syn), dc_, di)
            do, dj = dev_u64(words.shape), dev_u32((shots,))
            dec.decode_batch(shots, to_dev_u64(words), do, dj)
            cnt = dec.mc_run(p, 9 + seed, 17, 3000)
            return host_u64(dc_), host_u32(di), host_u64(do), host_u32(dj), cnt

        a, b = forms(run)
        same = all(np.array_equal(x, y) for x, y in zip(a, b))
        bad += not same
        print("%-32s p=%.2f cap=%3d: %s  (iterations/shot %.2f, failures %d of 3000)" % (name, p, cap, "same" if same else "DIFFER  <-- CHECK",
              float(a[4][4]) / 3000, int(a[4][1])), flush=True)
# CSS: hypergraph products of small regular codes
for (n, m, dv, dc) in [(20, 15, 3, 4), (12, 9, 3, 4), (16, 8, 2, 4)]:
    base = LinearCode.random_regular_code(n, m, dv, dc, seed=1 + seed)
    css = CssCode.hypergraph_product(base, base)
    for p, cap in ((0.01, 100), (0.04, 31)):
        a, b = forms(lambda: (CssDecoder(css, p, cap).mc_run(p, 5 + seed, 0, 20000),))
        same = np.array_equal(a[0], b[0])
        bad += not same
        print("css hgp(%d,%d) p=%.2f cap=%d: %s  %s" % (n, m, p, cap, "same" if same else "DIFFER  <-- CHECK", a[0].tolist()), flush=True)
print("differing cases:", bad)
