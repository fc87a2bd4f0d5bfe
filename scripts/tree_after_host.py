"""Reproducer: the tree-order sweep of one n = 4096 sequence (device-resident entry, caller's
stream) before and after ONE call of the host-buffer entry rnamc_bpp_batch in the same process."""
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
s4 = W.synthetic_seq(4096, 4096)
b4 = torch.from_numpy(s4).to(dev)
o4 = torch.empty(4096 * 4097 // 2, dtype=torch.float32, device=dev)
z4 = torch.empty(1, dtype=torch.float32, device=dev)
off4 = np.array([0, 4096], dtype=np.uint64)
oo4 = np.array([0, 4096 * 4097 // 2], dtype=np.uint64)
st = torch.cuda.current_stream().cuda_stream


def tree(label, reps=4):
    ctx.set("summation_mode", 1)
    ms = []
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.bpp_batch_device(1, b4.data_ptr(), off4, False, False, o4.data_ptr(), oo4, z4.data_ptr(), st)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    ctx.set("summation_mode", 0)
    print(f"{label}: tree n=4096 ms per call {[round(x, 1) for x in ms]}", flush=True)


what = sys.argv[1] if len(sys.argv) > 1 else "host"
if len(sys.argv) < 3 or sys.argv[2] != "nofirst":
    tree("fresh context")
if what == "host":
    mats, logz = ctx.bpp_batch([W.synthetic_seq(300, 7), W.synthetic_seq(200, 8)], False, False)
    tree("after one rnamc_bpp_batch call (2 short sequences)")
elif what.startswith("host") and what[4:].isdigit():
    cnt = int(what[4:])
    lens = W.batch_lengths(10000)[:cnt]
    seqs = [W.synthetic_seq(int(n), (10000 << 32) + k) for k, n in enumerate(lens)]
    for rep in range(3):
        t0 = time.perf_counter()
        mats, logz = ctx.bpp_batch(seqs, False, False)
        print(f"host entry, {cnt} sequences, call {rep}: {time.perf_counter() - t0:.2f} s", flush=True)
        del mats
    tree(f"after one rnamc_bpp_batch call ({cnt} sequences of the 10k batch)")
elif what == "device":
    s = W.synthetic_seq(300, 7)
    b = torch.from_numpy(s).to(dev)
    o = torch.empty(300 * 301 // 2, dtype=torch.float32, device=dev)
    z = torch.empty(1, dtype=torch.float32, device=dev)
    ctx.bpp_batch_device(1, b.data_ptr(), np.array([0, 300], dtype=np.uint64), False, False, o.data_ptr(),
                         np.array([0, 300 * 301 // 2], dtype=np.uint64), z.data_ptr(), st)
    torch.cuda.synchronize()
    tree("after one device-resident reference-order call")
