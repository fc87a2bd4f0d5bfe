"""Round-4 verdict item 1(a): how often is a WHOLE WAVE of the outside chains on the identity
piece of logsumexp (or folding -inf) in the same fold step?  Runs the 512 longest sequences of the
10k bench batch (one lock-step group) through the counting build of the library
(`make -C rna_algos_amd/csrc COUNT_FAR=1 OUT=../librnamc_count.so OBJDIR=build_count`, loaded through
RNAMC_LIB) and prints the tallies of FarCount (rnamc_kernels.hip).  Usage:
    RNAMC_LIB=$PWD/rna_algos_amd/librnamc_count.so python scripts/count_far.py [count] > profiles/r04_wave_uniform_far.txt
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rna_algos_amd import _lib, workloads as WL  # noqa: E402
from rna_algos_amd.utils import FoldScoreSets  # noqa: E402
from rna_algos_amd.mccaskill_algo import Context  # noqa: E402

NAMES = ["fold steps per wave", "  all lanes on the identity piece / -inf", "  all lanes' term -inf (no-op)",
         "lane-steps on the identity piece / -inf", "lane-steps with a -inf term", "k-steps per wave",
         "  all folds of the k-step wave-uniform far", "chunks (16 k pair tail / 8 k probs_multibranch)",
         "  all folds of the chunk wave-uniform far"]


def main():
    cnt = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    L = _lib.lib()
    if not hasattr(L, "rnamc_debug_far_counters"):
        raise SystemExit("not the counting build: set RNAMC_LIB to librnamc_count.so")
    L.rnamc_debug_far_counters.argtypes = [C.c_void_p, C.c_int]
    P = FoldScoreSets.synthetic(1)
    ctx = Context(P, device=0)
    ctx.set("group_max_seqs", cnt)
    lens = WL.batch_lengths(10000)
    order = np.argsort(-lens, kind="stable")[:cnt]
    seqs = [WL.synthetic_seq(int(lens[s]), (10000 << 32) + int(s)) for s in order]
    dev = torch.device("cuda:0")
    ln = np.array([len(s) for s in seqs], dtype=np.uint64)
    offsets = np.zeros(cnt + 1, dtype=np.uint64)
    np.cumsum(ln, out=offsets[1:])
    out_offsets = np.zeros(cnt + 1, dtype=np.uint64)
    np.cumsum(ln * (ln + 1) // 2, out=out_offsets[1:])
    bases = torch.from_numpy(np.concatenate(seqs)).to(dev)
    out = torch.empty(int(out_offsets[-1]), dtype=torch.float32, device=dev)
    logz = torch.empty(cnt, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    print(f"# {cnt} longest sequences of the 10k batch ({int(ln.min())}..{int(ln.max())} nt), one lock-step group, "
          f"synthetic tables seed 1; counting build (FarCount, rnamc_kernels.hip)")
    for contra in (False, True):
        buf = (C.c_uint64 * 32)()
        L.rnamc_debug_far_counters(None, 1)
        ctx.bpp_batch_device(cnt, bases.data_ptr(), offsets, contra, False, out.data_ptr(), out_offsets,
                             logz.data_ptr(), st)
        torch.cuda.synchronize()
        L.rnamc_debug_far_counters(buf, 0)
        v = np.array(list(buf), dtype=np.float64).reshape(2, 16)
        for role, name in ((0, "pair tail (k_outside<.,2>)"), (1, "probs_multibranch role (k_outside<.,5>)")):
            c = v[role]
            print(f"\n{'CONTRAfold' if contra else 'Turner'}, {name}")
            for x in range(9):
                print(f"  {NAMES[x]:<52s} {c[x]:.6e}")
            if c[0] > 0:
                print(f"  => wave-uniform far fold steps: {c[1] / c[0]:.4f} of all fold steps "
                      f"(all-(-inf) steps: {c[2] / c[0]:.4f}); per lane: {c[3] / (64 * c[0]):.4f} far, "
                      f"{c[4] / (64 * c[0]):.4f} -inf terms")
                print(f"  => k-steps with every fold wave-uniform far: {c[6] / max(c[5], 1):.4f}; "
                      f"whole chunks: {c[8] / max(c[7], 1):.4f}")
    ctx.close()


if __name__ == "__main__":
    main()
