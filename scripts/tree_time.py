"""Time the tree-order mode on one sequence (for rocprofv3 runs).  argv: n contra reps"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from rna_algos_amd.utils import FoldScoreSets
from rna_algos_amd.mccaskill_algo import Context
n = int(sys.argv[1]); contra = bool(int(sys.argv[2])); reps = int(sys.argv[3])
P = FoldScoreSets.synthetic(1)
ctx = Context(P, device=0)
ctx.set("summation_mode", 1)
ctx.set("profile", 1)
for k in sys.argv[4:]:
    name, v = k.split("=")
    ctx.set(name, int(v))
s = O.splitmix_seq(n, n)
for rep in range(reps):
    t0 = time.perf_counter()
    m, z = ctx.bpp_batch([s], contra, False)
    t1 = time.perf_counter()
    st = ctx.stats()
    print(f"tree n={n} contra={int(contra)} wall={1e3*(t1-t0):.1f} ms inside={st['ms_inside']:.2f} outside={st['ms_outside']:.2f} lnZ={float(z[0]):.4f}", flush=True)
