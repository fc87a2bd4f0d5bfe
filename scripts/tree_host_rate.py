"""Is the tree-order sweep's chain bound by the host's launch rate?  Device-resident entry with profile = 0:
the call returns when everything is ENQUEUED; wall to return = host time, wall to sync = GPU time.
(RNAMC_LIB=...librnamc_dbg.so and tree_debug=4 give empty kernels.)"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rna_algos_amd import workloads as W
from rna_algos_amd.utils import FoldScoreSets
from rna_algos_amd.mccaskill_algo import Context
P = FoldScoreSets.synthetic(1)
ctx = Context(P, device=0)
ctx.set("summation_mode", 1)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    ctx.set(k, int(v))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = W.synthetic_seq(n, n)
d = torch.device("cuda:0")
b = torch.from_numpy(np.ascontiguousarray(s)).to(d)
o = torch.empty(n * (n + 1) // 2, dtype=torch.float32, device=d)
z = torch.empty(1, dtype=torch.float32, device=d)
off = np.array([0, n], dtype=np.uint64)
oo = np.array([0, n * (n + 1) // 2], dtype=np.uint64)
for r in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.bpp_batch_device(1, b.data_ptr(), off, False, False, o.data_ptr(), oo, z.data_ptr(), 0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n} {' '.join(sys.argv[2:])}: enqueue returned after {1e3 * (t1 - t0):.2f} ms, GPU done after {1e3 * (t2 - t0):.2f} ms", flush=True)
