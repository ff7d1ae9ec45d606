"""Generates tests/golden/c1_vectors.npz with the CPU oracle (oracle/ldpc_oracle.c).

Run from the repo root:  python tests/golden/make_golden.py
The reference itself (Rust) cannot be run here, so these vectors come from the
oracle, which is pinned by the reference's known-answer tests
(tests/test_oracle_kats.py).  Inputs: the C1 configuration of BASELINE.json,
(3,4)-regular n=1200 code drawn by this library's sampler with code_seed=1,
BSC p=0.02, noise_seed=2024, shots 0..63; plus the HGP(20,15,3,4) CSS code of
examples/css_decoding.rs with depolarizing p=0.01 (p=0.05 for a harder set).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import ldpc_amd  # noqa: E402  (host-side code construction only, no GPU)
from oracle import oracle as orc  # noqa: E402

SEED = 2024
SHOTS = 64


def main():
    code = ldpc_amd.LinearCode.random_regular_code(1200, 900, 3, 4, seed=1)
    ptr, idx = code.csr()
    oc = orc.Code(1200, ptr=ptr.astype(np.int64), idx=idx.astype(np.uint32))
    errs, syns, flips_out, flips_n, bp_corr, bp_it = [], [], [], [], [], []
    for shot in range(SHOTS):
        e = orc.sample_bsc_words(SEED, shot, 0.02, 1200)
        s = oc.syndrome_words(e)
        o, _, nf = oc.flip_decode_words(e)
        c, it, _ = oc.bp_correction_for(0.02, 100, s)
        errs.append(e); syns.append(s); flips_out.append(o); flips_n.append(nf); bp_corr.append(c); bp_it.append(it)
    base = ldpc_amd.LinearCode.random_regular_code(20, 15, 3, 4, seed=1)
    css = ldpc_amd.CssCode.hypergraph_product(base, base)
    mats = [css.x_stabs_binary(), css.z_stabs_binary(), css.x_logicals_binary(), css.z_logicals_binary()]
    ocs = []
    css_arrays = {}
    for name, m in zip(("hx", "hz", "lx", "lz"), mats):
        p, i = m.csr()
        css_arrays["css_%s_ptr" % name] = p.astype(np.int64)
        css_arrays["css_%s_idx" % name] = i.astype(np.uint32)
        ocs.append(orc.Code(625, ptr=p.astype(np.int64), idx=i.astype(np.uint32)))
    css_counts = {}
    for p in (0.01, 0.05):
        css_counts["css_counters_p%g" % p] = orc.mc_css_bp(ocs[0], ocs[1], ocs[2], ocs[3], p, 100, SEED, 0, 2000,
                                                           nthreads=8)
    ex, ez = zip(*[orc.sample_depolarizing_words(SEED, s, 0.05, 625) for s in range(SHOTS)])
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "c1_vectors.npz"),
        h_ptr=ptr.astype(np.int64), h_idx=idx.astype(np.uint32),
        errors=np.array(errs), syndromes=np.array(syns), flip_out=np.array(flips_out),
        flip_count=np.array(flips_n, dtype=np.uint32), bp_corr=np.array(bp_corr), bp_iters=np.array(bp_it, dtype=np.uint32),
        c1_flip_counters_10000=oc.mc_flip(0.02, SEED, 0, 10000, nthreads=8),
        c1_bp_counters_2000=oc.mc_bp(0.02, 100, SEED, 0, 2000, nthreads=8),
        depol_x=np.array(ex), depol_z=np.array(ez),
        **css_arrays, **css_counts)
    print("written")


if __name__ == "__main__":
    main()
