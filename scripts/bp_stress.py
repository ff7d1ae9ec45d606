"""One-off parity stress of the BP kernels against the oracle on assorted codes (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd
import numpy as np
from ldpc_amd import BpDecoder, LinearCode, nwords
from oracle import oracle as orc
from tests.gpu_util import dev_f64, dev_u32, dev_u64, host_u32, host_u64, oracle_code, random_sparse_rows, to_dev_u64

SEED_SHIFT = int(os.environ.get("LDPC_STRESS_SEED", "0"))   # other codes and noise per value
rng = np.random.default_rng(11 + SEED_SHIFT)
cases = []
for (n, m, dv, dc) in [(2000, 1000, 5, 10), (2400, 1200, 4, 8), (1500, 900, 3, 5), (3000, 2500, 5, 6), (640, 320, 3, 6), (6400, 3200, 3, 6)]:
    cases.append(("reg(%d,%d) n=%d" % (dv, dc, n), LinearCode.random_regular_code(n, m, dv, dc, seed=3 + SEED_SHIFT)))
for (n, m, maxdeg) in [(2000, 1200, 10), (900, 700, 14), (5000, 2000, 5)]:
    cases.append(("irregular n=%d m=%d deg<=%d" % (n, m, maxdeg), LinearCode.from_parity_check_matrix(n, random_sparse_rows(rng, n, m, maxdeg))))
bad = 0
for name, code in cases:
    n, m = code.len(), code.num_checks()
    oc = oracle_code(code)
    for path in ("persistent", "stream"):
        os.environ["LDPC_HIP_BP_PATH"] = path
        ldpc_amd.reload_switches()
        for p, cap in ((0.01, 40), (0.06, 15)):
            shots = int(sys.argv[1]) if len(sys.argv) > 1 else 96
            syn = np.zeros((shots, max(nwords(m), 1)), dtype=np.uint64)
            for i in range(shots):
                syn[i, : nwords(m)] = oc.syndrome_words(orc.sample_bsc_words(4 + SEED_SHIFT, i, p, n))
            dec = BpDecoder(code, p, cap)
            dc_, di, dp = dev_u64((shots, nwords(n))), dev_u32((shots,)), dev_f64((shots, n))
            dec.correction_for_batch(shots, to_dev_u64(syn), dc_, di, dp)
            gc, gi, gp = host_u64(dc_), host_u32(di), dp.cpu().numpy()
            same = it_same = 0
            worst = 0.0
            for i in range(shots):
                c, it, post = oc.bp_correction_for(p, cap, syn[i, : nwords(m)])
                eq = (gc[i] == c).all()
                same += eq
                it_same += gi[i] == it
                if eq:
                    fin = np.isfinite(post) & (post != 0)
                    if fin.any():
                        worst = max(worst, float(np.max(np.abs(gp[i][fin] - post[fin]) / np.abs(post[fin]))))
            flag = "" if same >= shots - 1 and worst < 1e-5 else "  <-- CHECK"
            bad += bool(flag)
            print("%-34s %-10s p=%.2f cap=%d: corrections equal %d/%d, iterations equal %d/%d, worst LLR rel err %.1e%s" %
                  (name, path, p, cap, same, shots, it_same, shots, worst, flag), flush=True)
print("suspicious cases:", bad)
