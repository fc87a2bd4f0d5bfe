"""Would two groups of the tree-order batch form side by side fill the launch gaps?  Two contexts on one device,
each with half of the slice (alternating in length order), two host threads, two streams — against one context
with the whole slice.  usage: tree_two_ctx_batch.py [count] [stride]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rna_algos_amd import workloads as W
from rna_algos_amd.utils import FoldScoreSets
from rna_algos_amd.mccaskill_algo import Context

count = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 10
seqs = sorted(W.batch(10000)[::stride][:count], key=len, reverse=True)
dev = torch.device("cuda:0")
P = FoldScoreSets.synthetic(1)


def prep(ss):
    lens = np.array([len(s) for s in ss], dtype=np.uint64)
    off = np.zeros(len(ss) + 1, dtype=np.uint64); np.cumsum(lens, out=off[1:])
    oo = np.zeros(len(ss) + 1, dtype=np.uint64); np.cumsum(lens * (lens + np.uint64(1)) // np.uint64(2), out=oo[1:])
    return dict(n=len(ss), nt=int(lens.sum()), off=off, oo=oo, b=torch.from_numpy(np.concatenate(ss)).to(dev),
                o=torch.empty(int(oo[-1]), dtype=torch.float32, device=dev),
                z=torch.empty(len(ss), dtype=torch.float32, device=dev), st=torch.cuda.Stream(device=dev))


def run(c, d):
    c.bpp_batch_device(d["n"], d["b"].data_ptr(), d["off"], False, False, d["o"].data_ptr(), d["oo"], d["z"].data_ptr(),
                       d["st"].cuda_stream)


whole = prep(seqs)
c0 = Context(P, device=0); c0.set("summation_mode", 1)
for ws in (128, 64):
    c0.set("group_ws_bytes", ws << 30)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(c0, whole); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"one context, groups of {ws} GB: {best * 1e3:.1f} ms  {whole['nt'] / best / 1e3:.1f} k nt/s", flush=True)
c0.close()
halves = [prep(seqs[0::2]), prep(seqs[1::2])]
cs = []
for _ in range(2):
    c = Context(P, device=0); c.set("summation_mode", 1); c.set("group_ws_bytes", 64 << 30); cs.append(c)
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(c, d)) for c, d in zip(cs, halves)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print(f"two contexts side by side, 64 GB each: {best * 1e3:.1f} ms  {whole['nt'] / best / 1e3:.1f} k nt/s", flush=True)
