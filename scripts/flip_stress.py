"""One-off parity stress of the flip kernels against the oracle on assorted codes (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldpc_amd
import numpy as np
from ldpc_amd import FlipDecoder, LinearCode, nwords
from oracle import oracle as orc
from tests.gpu_util import dev_u32, dev_u64, host_u32, host_u64, oracle_code, random_sparse_rows

SEED_SHIFT = int(os.environ.get("LDPC_STRESS_SEED", "0"))   # other codes and noise per value
rng = np.random.default_rng(12 + SEED_SHIFT)
cases = []
for (n, m, dv, dc) in [(2000, 1000, 5, 10), (2400, 1200, 4, 8), (1500, 900, 3, 5), (3000, 2500, 5, 6), (640, 320, 3, 6),
                       (6400, 3200, 3, 6), (2100, 900, 3, 7), (1400, 1200, 6, 7), (2000, 1750, 7, 8), (30000, 15000, 3, 6),
                       (1000, 800, 8, 10)]:
    cases.append(("reg(%d,%d) n=%d" % (dv, dc, n), LinearCode.random_regular_code(n, m, dv, dc, seed=3 + SEED_SHIFT)))
for (n, m, maxdeg) in [(2000, 1200, 10), (900, 700, 14), (5000, 2000, 5), (20000, 9000, 6)]:
    cases.append(("irregular n=%d m=%d deg<=%d" % (n, m, maxdeg), LinearCode.from_parity_check_matrix(n, random_sparse_rows(rng, n, m, maxdeg))))
bad = 0
for name, code in cases:
    n, m = code.len(), code.num_checks()
    oc = oracle_code(code)
    for lanes in ("16", "8"):
        os.environ["LDPC_HIP_FLIP_LANES"] = lanes
        ldpc_amd.reload_switches()
        dec = FlipDecoder(code)
        for p in (0.004, 0.03, 0.3):
            shots = 64
            de, ds, do = dev_u64((shots, nwords(n))), dev_u64((shots, max(nwords(m), 1))), dev_u64((shots, nwords(n)))
            cnt = dec.mc_run(p, 6 + SEED_SHIFT, 1000, shots, de, ds, do)
            ge, gs, go = host_u64(de), host_u64(ds), host_u64(do)
            ok = flips = 0
            for i in range(shots):
                e = orc.sample_bsc_words(6 + SEED_SHIFT, 1000 + i, p, n)
                o, _, nf = oc.flip_decode_words(e)
                ok += bool((ge[i] == e).all() and (gs[i][: nwords(m)] == oc.syndrome_words(e)).all() and (go[i] == o).all())
                flips += nf
            flag = "" if ok == shots and flips == int(cnt[4]) else "  <-- CHECK"
            bad += bool(flag)
            print("%-34s lanes=%-2s p=%.3f: shots equal %d/%d, flips %d vs %d%s" % (name, lanes, p, ok, shots, int(cnt[4]), flips, flag), flush=True)
print("suspicious cases:", bad)
