"""Tree-order mode on a slice of the bench batch, device-resident, under sets of context knobs:
one line per set with the pass time and the sweeps' device times (events of the `profile` knob).
With RNAMC_LIB=.../librnamc_dbg.so (make DEBUG_KNOBS=1) the tree_debug bits switch stages off
(1: generic 2-loops, 2: near-band cubic terms, 32: mid-field kernels, 64: sums_external) — results
are wrong then, the times say what a stage costs.
usage: tree_batch_knobs.py <count> <contra 0|1> <reps> [k=v,k=v ...] (a lone '-' = no knobs)
       env TREE_STRIDE=s takes every s-th sequence of the 10k batch instead of the first <count>"""
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


def main():
    count, contra, reps = int(sys.argv[1]), sys.argv[2] == "1", int(sys.argv[3])
    sets = sys.argv[4:] or ["-"]
    stride = int(os.environ.get("TREE_STRIDE", "0"))
    if stride:
        seqs = W.batch(10000)[::stride][:count]
    else:
        seqs = W.batch(count)
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    out_offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens * (lens + np.uint64(1)) // np.uint64(2), out=out_offsets[1:])
    dev = torch.device("cuda:0")
    d_bases = torch.from_numpy(np.concatenate(seqs)).to(dev)
    d_out = torch.empty(int(out_offsets[-1]), dtype=torch.float32, device=dev)
    d_logz = torch.empty(len(seqs), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    nt = int(lens.sum())
    print(f"{len(seqs)} sequences, {nt} nt, longest {int(lens.max())}, contra={int(contra)}", flush=True)
    for ks in sets:
        ctx = Context(FoldScoreSets.synthetic(1), device=0)
        ctx.set("summation_mode", 1)
        ctx.set("profile", 1)
        if ks != "-":
            for kv in ks.split(","):
                k, v = kv.split("=")
                ctx.set(k, int(v))
        best = None
        for _ in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.bpp_batch_device(len(seqs), d_bases.data_ptr(), offsets, contra, False, d_out.data_ptr(),
                                 out_offsets, d_logz.data_ptr(), stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        st = ctx.stats()
        z = d_logz.float().cpu().numpy()
        print(f"{ks:40s} pass {best * 1e3:9.1f} ms  {nt / best / 1e3:8.1f} k nt/s   inside {st['ms_inside']:9.1f}  "
              f"outside {st['ms_outside']:9.1f}  other {st['ms_other']:7.1f}   lnZ[0] {z[0]:.4f} lnZ[-1] {z[-1]:.4f}",
              flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
