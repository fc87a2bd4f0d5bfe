"""Side-stream check with TWO live contexts on one device: A runs a host-entry batch and stays
alive, then B is created and runs tree-order calls (device-resident, null stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from rna_algos_amd import workloads as W
from rna_algos_amd.utils import FoldScoreSets
from rna_algos_amd.mccaskill_algo import Context

P = FoldScoreSets.synthetic(1)
dev = torch.device("cuda:0")
n = 2048
s = W.synthetic_seq(n, n)
b = torch.from_numpy(s).to(dev)
o = torch.empty(n * (n + 1) // 2, dtype=torch.float32, device=dev)
z = torch.empty(1, dtype=torch.float32, device=dev)
off = np.array([0, n], dtype=np.uint64)
oo = np.array([0, n * (n + 1) // 2], dtype=np.uint64)


def tree(c, band):
    c.set("summation_mode", 1); c.set("tree_band", band)
    ms = []
    for r in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c.bpp_batch_device(1, b.data_ptr(), off, False, False, o.data_ptr(), oo, z.data_ptr(), 0)
        torch.cuda.synchronize()
        if r: ms.append(round((time.perf_counter() - t0) * 1e3, 1))
    c.set("summation_mode", 0); c.set("tree_band", 64)
    return ms


rng = np.random.default_rng(31)
batch = [rng.integers(0, 4, int(k)).astype(np.uint8) for k in rng.integers(300, 600, 300)]
ctxs = []
for k in range(3):
    c = Context(P, device=0)
    c.bpp_batch(batch, False, False)
    ctxs.append(c)
    print(f"context {k} (created after {k} live contexts that ran host-entry batches): banded {tree(c, 64)}, unbanded {tree(c, 0)}", flush=True)
print("context 0 again: banded", tree(ctxs[0], 64), flush=True)
