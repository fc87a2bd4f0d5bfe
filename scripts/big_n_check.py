"""One-off sanity check at lengths no oracle run can pin (n = 8192, 12001): key set,
range and row sums of the base-pairing probabilities.  Run on the GPU box."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rna_algos_amd.utils import FoldScoreSets
from rna_algos_amd.mccaskill_algo import Context
from rna_algos_amd import workloads as W
P = FoldScoreSets.synthetic(1)
ctx = Context(P, device=0)
for n in (8192, 12001):
    s = W.synthetic_seq(n, n)
    t = time.time()
    mats, logz = ctx.bpp_batch([s], False, False)
    dt = time.time() - t
    m = mats[0].packed
    pres = m >= -0.5
    vals = m[pres]
    off = 0
    rowsum = np.zeros(n)
    for d in range(n):
        row = m[off:off + n - d]
        a, b = s[:n - d].astype(int), s[d:].astype(int)
        canon = ((a + b == 3) | (a + b == 5)) & (d >= 4)
        assert np.array_equal(row >= -0.5, canon), d
        r = np.where(canon, row, 0.0)
        rowsum[:n - d] += r
        rowsum[d:] += r
        off += n - d
    print(n, f"{dt:.1f}s", float(logz[0]), float(vals.min()), float(vals.max()), float(rowsum.max()), flush=True)
