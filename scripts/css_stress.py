"""Parity stress of the CSS Monte-Carlo path against the oracle on assorted hypergraph-product codes (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ldpc_amd import CNT_FAILURES, CNT_SHOTS, CNT_WORK, CNT_XFAIL, CNT_ZFAIL, CssCode, CssDecoder, LinearCode
from oracle import oracle as orc

SEED_SHIFT = int(os.environ.get("LDPC_STRESS_SEED", "0"))
bad = 0
for (n, m, dv, dc), (n2, m2, dv2, dc2) in [((12, 9, 3, 4), (12, 9, 3, 4)), ((16, 12, 3, 4), (12, 8, 2, 3)), ((20, 15, 3, 4), (20, 15, 3, 4)),
                                           ((15, 10, 2, 3), (16, 12, 3, 4)), ((24, 12, 3, 6), (12, 9, 3, 4))]:
    a = LinearCode.random_regular_code(n, m, dv, dc, seed=1 + SEED_SHIFT)
    b = LinearCode.random_regular_code(n2, m2, dv2, dc2, seed=2 + SEED_SHIFT)
    css = CssCode.hypergraph_product(a, b)
    d = {"hx": css.x_stabs_binary(), "hz": css.z_stabs_binary(), "lx": css.x_logicals_binary(), "lz": css.z_logicals_binary()}
    oc = {k: orc.Code(css.len(), rows=v.parity_check_matrix()) for k, v in d.items()}
    for p, cap, shots in ((0.01, 60, 3000), (0.05, 25, 1500), (0.01, 30, 12000)):
        cnt = CssDecoder(css, p, cap).mc_run(p, 7 + SEED_SHIFT, 100, shots)
        ref = orc.mc_css_bp(oc["hx"], oc["hz"], oc["lx"], oc["lz"], p, cap, 7 + SEED_SHIFT, 100, shots, nthreads=16)
        tol = max(2, int(round(1e-4 * shots)))
        ok = cnt[CNT_SHOTS] == shots and all(abs(int(cnt[k]) - int(ref[k])) <= tol for k in (CNT_FAILURES, CNT_XFAIL, CNT_ZFAIL)) \
            and abs(int(cnt[CNT_WORK]) - int(ref[CNT_WORK])) <= 200 * tol
        bad += not ok
        print("hgp(%d,%d)x(%d,%d) %4d qubits p=%.2f cap=%d shots=%d: failures %d vs %d, x %d vs %d, z %d vs %d, work %d vs %d %s"
              % (n, m, n2, m2, css.len(), p, cap, shots, cnt[CNT_FAILURES], ref[CNT_FAILURES], cnt[CNT_XFAIL], ref[CNT_XFAIL],
                 cnt[CNT_ZFAIL], ref[CNT_ZFAIL], cnt[CNT_WORK], ref[CNT_WORK], "" if ok else "  <-- CHECK"), flush=True)
print("suspicious cases:", bad)
