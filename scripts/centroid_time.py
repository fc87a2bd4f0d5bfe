"""Time the gamma-centroid folds of one bpp matrix: host fold per threshold against the GPU fill
for all thresholds at once.  argv: n [n_gammas]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from rna_algos_amd.utils import FoldScoreSets
from rna_algos_amd.mccaskill_algo import Context
from rna_algos_amd.centroid_fold import centroid_fold, centroid_fold_multi
n = int(sys.argv[1]); ng = int(sys.argv[2]) if len(sys.argv) > 2 else 18
P = FoldScoreSets.synthetic(1)
ctx = Context(P, device=0)
ctx.set("summation_mode", 1)  # (bpp source only; tree order is the fast one on a lone sequence)
s = O.splitmix_seq(n, n)
m, _ = ctx.bpp_batch([s], False, False)
gammas = [2.0 ** k for k in range(-7, 11)][:ng]
for rep in range(2):
    t0 = time.perf_counter(); folds = centroid_fold_multi(ctx, m[0], n, gammas); t1 = time.perf_counter()
    print(f"GPU fill + host traceback, n={n}, {len(gammas)} thresholds: {1e3*(t1-t0):.1f} ms", flush=True)
t0 = time.perf_counter(); h = centroid_fold(m[0], n, gammas[len(gammas) // 2]); t1 = time.perf_counter()
print(f"host fold (rnamc_centroid_fold), ONE threshold: {1e3*(t1-t0):.1f} ms  (x{len(gammas)} thresholds = {1e3*(t1-t0)*len(gammas):.0f} ms on one core)")
g = folds[len(gammas) // 2]
print("identical:", g.basepair_pos_pairs == h.basepair_pos_pairs and np.float32(g.expect_accuracy) == np.float32(h.expect_accuracy), "pairs:", len(g.basepair_pos_pairs))
