"""Statistics of the serial flip process (flip.rs:21-43) that the flip kernel's design rests on (CPU, diagnostics):
how often the bit flipped next is the second-lowest flippable bit seen before the flip (row prefetch hit rate),
how many windows of 512 mask bits a scan crosses, flips per shot."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ldpc_amd import LinearCode
from oracle import oracle as orc

n, m, dv, dc, p = 10000, 5000, 3, 6, float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
shots = int(sys.argv[2]) if len(sys.argv) > 2 else 20
code = LinearCode.random_regular_code(n, m, dv, dc, seed=1)
ptr, idx = code.csr()
rows = [idx[ptr[c]:ptr[c + 1]].tolist() for c in range(m)]
cols = [[] for _ in range(n)]
for c, r in enumerate(rows):
    for b in r:
        cols[b].append(c)
nbr = []
for b in range(n):
    s = set()
    for c in cols[b]:
        s.update(rows[c])
    nbr.append(sorted(s))

tot_flips = hits = hits_win = same_win = 0
new_lower = 0
flips_list = []
for shot in range(shots):
    words = orc.sample_bsc_words(2024, shot, p, n)
    e = np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(np.int8)
    S = np.zeros(m, dtype=np.int8)
    for b in np.nonzero(e)[0]:
        for c in cols[b]:
            S[c] ^= 1
    unsat = np.array([sum(S[c] for c in cols[b]) for b in range(n)])
    F = set(np.nonzero(unsat * 2 > dv)[0].tolist())
    import heapq
    nfl = 0
    pred = None
    while F:
        srt = sorted(F)[:2]
        b = srt[0]
        second = srt[1] if len(srt) > 1 else None
        if pred is not None:
            if pred == b:
                hits += 1
                if pred_same_window:
                    hits_win += 1
        pred = second
        pred_same_window = second is not None and (second >> 9) == (b >> 9)
        if pred_same_window:
            same_win += 1
        for c in cols[b]:
            S[c] ^= 1
        lower = False
        for x in nbr[b]:
            u = sum(S[c] for c in cols[x])
            fl = u * 2 > len(cols[x])
            if fl and x not in F:
                F.add(x)
                if second is None or x < second:
                    lower = True
            elif not fl and x in F:
                F.discard(x)
        new_lower += lower
        nfl += 1
    tot_flips += nfl
    flips_list.append(nfl)
print("p=%g shots=%d flips/shot=%.1f (min %d max %d sd %.1f)" % (p, shots, tot_flips / shots, min(flips_list), max(flips_list),
                                                          np.std(flips_list)))
print("next flip == second-lowest before the flip: %.3f" % (hits / tot_flips))
print("  ... and that second-lowest was in the same 512-bit window: pred-available %.3f, hit %.3f" % (same_win / tot_flips, hits_win / tot_flips))
print("a newly flippable neighbour below the second-lowest: %.3f of flips" % (new_lower / tot_flips))
fl = np.array(flips_list)
for g in (4, 8, 16):
    k = len(fl) // g * g
    print("mean of max over %2d shots / mean = %.3f" % (g, fl[:k].reshape(-1, g).max(axis=1).mean() / fl.mean()))
