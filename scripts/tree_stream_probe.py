"""Does the caller's stream matter for the tree-order sweep?  null stream vs a non-blocking one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from rna_algos_amd import workloads as W
from rna_algos_amd.utils import FoldScoreSets
from rna_algos_amd.mccaskill_algo import Context

P = FoldScoreSets.synthetic(1)
ctx = Context(P, device=0)
ctx.set("profile", 1)
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s4 = W.synthetic_seq(n, n)
b4 = torch.from_numpy(s4).to(dev)
o4 = torch.empty(n * (n + 1) // 2, dtype=torch.float32, device=dev)
z4 = torch.empty(1, dtype=torch.float32, device=dev)
off4 = np.array([0, n], dtype=np.uint64)
oo4 = np.array([0, n * (n + 1) // 2], dtype=np.uint64)
ctx.set("summation_mode", 1)
side = torch.cuda.Stream(device=dev)
hi = torch.cuda.Stream(device=dev, priority=-1)
for label, st in (("null stream", 0), ("torch.cuda.Stream()", side.cuda_stream),
                  ("torch.cuda.Stream(priority=-1)", hi.cuda_stream), ("null stream", 0)):
    ms = []
    for r in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.bpp_batch_device(1, b4.data_ptr(), off4, False, False, o4.data_ptr(), oo4, z4.data_ptr(), st)
        torch.cuda.synchronize()
        ms.append(round((time.perf_counter() - t0) * 1e3, 2))
    print(f"{label}: {ms}", flush=True)
