"""Scratch probe of the tree-order mode on the GPU: deviations from the f64 evaluation of the
same recurrences and from the reference-order mode, and timings.  (Not a test.)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from rna_algos_amd.utils import FoldScoreSets, read_fasta
from rna_algos_amd.mccaskill_algo import Context

P = FoldScoreSets.synthetic(1)
ctx = Context(P, device=0)

def run(seqs, contra, short, mode):
    ctx.set("summation_mode", mode)
    return ctx.bpp_batch(seqs, contra, short)

def cmp(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    ka, kb = a >= -0.5, b >= -0.5
    same = bool(np.array_equal(ka, kb))
    both = ka & kb
    return same, float(np.max(np.abs(a[both] - b[both]))) if both.any() else 0.0

seqs = [r[1] for r in read_fasta(os.path.join(ROOT, "tests", "golden", "sampled_trnas.fa"))]
for n, seed in [(1, 1), (3, 2), (4, 3), (5, 4), (7, 5), (12, 6), (33, 7), (64, 8), (100, 9), (257, 10)]:
    seqs.append(O.splitmix_seq(n, seed))
for contra, short in [(False, False), (True, False), (True, True)]:
    mt, zt = run(seqs, contra, short, 1)
    mr, zr = run(seqs, contra, short, 0)
    for s, a, r, za, zrr in zip(seqs, mt, mr, zt, zr):
        xb, xz = O.exact_bpp(P.ptr, s, contra, short)
        k1, d1 = cmp(a.packed, xb)
        k2, d2 = cmp(r.packed, xb)
        print(f"contra={int(contra)} short={int(short)} n={len(s):4d} keys_tree_vs_exact={k1} "
              f"dp_tree={d1:.2e} dp_ref={d2:.2e} dlnZ_tree={float(za)-xz:+.2e} dlnZ_ref={float(zrr)-xz:+.2e}", flush=True)
# one-by-one vs batch identical?
mt1, zt1 = run(seqs, False, False, 1)
ok = True
for x, s in enumerate(seqs):
    m1, z1 = run([s], False, False, 1)
    ok &= np.array_equal(m1[0].packed, mt1[x].packed) and z1[0] == zt1[x]
print("lone == batch (tree):", ok)
for n, contra in [(1024, True), (1024, False), (2048, False), (4096, False), (4096, True)]:
    s = O.splitmix_seq(n, n)
    ctx.set("profile", 1)
    for rep in range(3):
        t0 = time.perf_counter()
        mt, zt = run([s], contra, False, 1)
        t1 = time.perf_counter()
        st = ctx.stats()
        print(f"tree n={n} contra={int(contra)} wall={1e3*(t1-t0):.1f} ms inside={st['ms_inside']:.2f} "
              f"outside={st['ms_outside']:.2f} other={st['ms_other']:.2f} lnZ={float(zt[0]):.4f}", flush=True)
    mt2, zt2 = run([s], contra, False, 1)
    print("  deterministic:", np.array_equal(mt[0].packed, mt2[0].packed) and zt[0] == zt2[0])
    t0 = time.perf_counter()
    mr, zr = run([s], contra, False, 0)
    t1 = time.perf_counter()
    k, d = cmp(mt[0].packed, mr[0].packed)
    print(f"  ref-order wall={1e3*(t1-t0):.1f} ms lnZ={float(zr[0]):.4f}; keys_equal={k} max|dp|={d:.3e} dlnZ={float(zt[0])-float(zr[0]):+.3e}", flush=True)
    if n <= 1024:
        xb, xz = O.exact_bpp(P.ptr, s, contra, False)
        print(f"  vs exact f64: tree dp={cmp(mt[0].packed, xb)[1]:.3e} dlnZ={float(zt[0])-xz:+.3e}; "
              f"ref dp={cmp(mr[0].packed, xb)[1]:.3e} dlnZ={float(zr[0])-xz:+.3e}", flush=True)
