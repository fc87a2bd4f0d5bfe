"""Timing breakdown of the band launches (debug-knob build: make DEBUG_KNOBS=1 OUT=../librnamc_dbg.so
OBJDIR=build_dbg; RNAMC_LIB=.../librnamc_dbg.so): n = 4096 tree-order sweeps with parts of the band
kernels switched off (results wrong, timing only)."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = W.synthetic_seq(n, n)
d = torch.device("cuda:0")
b = torch.from_numpy(np.ascontiguousarray(s)).to(d)
o = torch.empty(n * (n + 1) // 2, dtype=torch.float32, device=d)
z = torch.empty(1, dtype=torch.float32, device=d)
off = np.array([0, n], dtype=np.uint64)
oo = np.array([0, n * (n + 1) // 2], dtype=np.uint64)
for name, dbg in (("all", 0), ("no probes", 128), ("no edges", 256), ("no probes, no edges", 384), ("no phase B", 512),
                  ("phase B only", 384), ("prologue only (no A, no B)", 896), ("mid / ext kernels off too", 896 + 32 + 64)):
    ctx.set("tree_debug", dbg)
    ms = []
    for r in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.bpp_batch_device(1, b.data_ptr(), off, False, False, o.data_ptr(), oo, z.data_ptr(), 0)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    st = ctx.stats()
    print(f"{name:34s}: {np.median(ms[1:]):7.2f} ms  inside {st['ms_inside']:6.2f} outside {st['ms_outside']:6.2f}  "
          f"launches {st['launches_inside']}+{st['launches_outside']}", flush=True)
