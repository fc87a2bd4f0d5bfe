"""ms per sequence of lone sequences in tree-order mode (device-resident, median of 5 after a warm-up).
Usage: python scripts/tree_ms.py [n ...]   (RNAMC_LIB picks the library build)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rna_algos_amd import workloads as W  # noqa: E402
from rna_algos_amd.utils import FoldScoreSets  # noqa: E402
from rna_algos_amd.mccaskill_algo import Context  # noqa: E402

P = FoldScoreSets.synthetic(1)
ctx = Context(P, device=0)
ctx.set("summation_mode", 1)
ctx.set("profile", 1)
for kv in filter(None, os.environ.get("SETS", "").split(",")):
    k, v = kv.split("=")
    ctx.set(k, int(v))
d = torch.device("cuda:0")
for n in [int(x) for x in (sys.argv[1:] or ["1024", "4096"])]:
    s = W.synthetic_seq(n, n)
    b = torch.from_numpy(np.ascontiguousarray(s)).to(d)
    o = torch.empty(n * (n + 1) // 2, dtype=torch.float32, device=d)
    z = torch.empty(1, dtype=torch.float32, device=d)
    off = np.array([0, n], dtype=np.uint64)
    oo = np.array([0, n * (n + 1) // 2], dtype=np.uint64)
    for contra in (False, True):
        ms = []
        for r in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.bpp_batch_device(1, b.data_ptr(), off, contra, False, o.data_ptr(), oo, z.data_ptr(), 0)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
        st = ctx.stats()
        print(f"{os.path.basename(os.environ.get('RNAMC_LIB', 'librnamc.so'))} n={n} contra={contra}: "
              f"{np.median(ms[1:]):.2f} ms (calls {[round(x, 1) for x in ms[1:]]}) inside {st['ms_inside']:.2f} "
              f"outside {st['ms_outside']:.2f} lnZ {float(z[0]):.4f}", flush=True)
ctx.close()
