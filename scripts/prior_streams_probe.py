"""Do streams that exist BEFORE the context is created (an application's own, RCCL's) change the
sweeps?  argv[1] = number of torch streams created and used first (half of them high priority)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from rna_algos_amd import workloads as W
from rna_algos_amd.utils import FoldScoreSets
from rna_algos_amd.mccaskill_algo import Context

k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda:0")
keep = []
for x in range(k):
    st = torch.cuda.Stream(device=dev, priority=-1 if x % 2 else 0)
    with torch.cuda.stream(st):
        keep.append(torch.zeros(1024, device=dev) + x)
    keep.append(st)
torch.cuda.synchronize()
P = FoldScoreSets.synthetic(1)
ctx = Context(P, device=0)
ctx.set("profile", 1)
lens = W.batch_lengths(10000)
order = np.argsort(-lens, kind="stable")[:256]
seqs = [W.synthetic_seq(int(lens[s]), (10000 << 32) + int(s)) for s in order]
ctx.set("group_max_seqs", 256)
ln = np.array([len(s) for s in seqs], dtype=np.uint64)
offsets = np.zeros(len(seqs) + 1, dtype=np.uint64); np.cumsum(ln, out=offsets[1:])
oo = np.zeros(len(seqs) + 1, dtype=np.uint64); np.cumsum(ln * (ln + 1) // 2, out=oo[1:])
bases = torch.from_numpy(np.concatenate(seqs)).to(dev)
out = torch.empty(int(oo[-1]), dtype=torch.float32, device=dev)
logz = torch.empty(len(seqs), dtype=torch.float32, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.bpp_batch_device(len(seqs), bases.data_ptr(), offsets, False, False, out.data_ptr(), oo, logz.data_ptr(), 0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = ctx.stats()
    print(f"{k} prior streams: top256 rep{rep} {dt*1e3:.0f} ms inside={s['ms_inside']:.0f} outside={s['ms_outside']:.0f}", flush=True)
n = 2048
s1 = W.synthetic_seq(n, n)
b = torch.from_numpy(s1).to(dev)
o = torch.empty(n * (n + 1) // 2, dtype=torch.float32, device=dev)
z = torch.empty(1, dtype=torch.float32, device=dev)
off1 = np.array([0, n], dtype=np.uint64); oo1 = np.array([0, n * (n + 1) // 2], dtype=np.uint64)
for band in (64, 0):
    ctx.set("summation_mode", 1); ctx.set("tree_band", band)
    ms = []
    for r in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.bpp_batch_device(1, b.data_ptr(), off1, False, False, o.data_ptr(), oo1, z.data_ptr(), 0)
        torch.cuda.synchronize()
        if r: ms.append(round((time.perf_counter() - t0) * 1e3, 1))
    ctx.set("summation_mode", 0); ctx.set("tree_band", 64)
    print(f"{k} prior streams: tree n=2048 band {band}: {ms}", flush=True)
