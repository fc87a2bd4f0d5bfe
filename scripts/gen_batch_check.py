import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from rna_algos_amd import workloads as W
from rna_algos_amd.utils import FoldScoreSets
from rna_algos_amd.mccaskill_algo import Context
P = FoldScoreSets.synthetic(1)
ctx = Context(P, device=0)
ctx.set("summation_mode", 1)
for n in (257, 410, 900):
    s = W.synthetic_seq(n, 13 * n + 5)
    ctx.set("tree_lane", 0)
    CT = bool(int(os.environ.get("CT", "0"))); m0, z0 = ctx.bpp_batch([s], CT, False)
    for gb in (1, 2, 3):
        ctx.set("tree_lane", 2); ctx.set("tree_gen_batch", gb)
        m, z = ctx.bpp_batch([s], CT, False)
        a = np.asarray(m[0].packed, np.float64); b = np.asarray(m0[0].packed, np.float64)
        both = (a >= -0.5) & (b >= -0.5)
        print(n, gb, float(z[0]) - float(z0[0]), np.abs(a[both] - b[both]).max(), flush=True)
