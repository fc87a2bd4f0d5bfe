"""Experiment (debug build with -DRNAMC_GEN_DIAGS=4 -DRNAMC_DEBUG_KNOBS): where do the generic 2-loop sums of FOUR
diagonals a launch differ from three?  Dumps T_GEN_D and the X4 planes of one sequence after the inside sweep's
values are final (the outside sweep overwrites both: tree_debug... so the run is compared through sums_close instead:
T_QB_D, slot 31, which the outside sweep leaves alone)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rna_algos_amd import workloads as W
from rna_algos_amd.utils import FoldScoreSets
from rna_algos_amd.mccaskill_algo import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
s = W.synthetic_seq(n, 13 * n + 5)
P = FoldScoreSets.synthetic(1)
res = {}
for gb in (3, 4):
    ctx = Context(P, device=0)
    ctx.set("summation_mode", 1); ctx.set("tree_lane", 2); ctx.set("tree_gen_batch", gb)
    path = f"/tmp/qb_{gb}.bin"
    os.environ["RNAMC_DUMP_SLOT"] = f"31,1,{path}"
    ctx.bpp_batch([s], False, False)
    raw = np.fromfile(path, dtype=np.uint8)
    hdr = raw[:24].view(np.uint64); nn, ld, msz = (int(x) for x in hdr)
    res[gb] = raw[24:].view(np.float32)[: nn * ld].reshape(nn, ld)
    ctx.close()
a, b = res[3], res[4]
print("n", nn, "ld", ld)
for d in range(nn):
    ra, rb = a[d, : nn - d], b[d, : nn - d]
    fin = np.isfinite(ra) | np.isfinite(rb)
    diff = np.where(fin & (ra != rb))[0]
    if len(diff):
        i = int(diff[0])
        print(f"first differing diagonal d={d}: {len(diff)} cells differ, e.g. i={i}: three {ra[i]!r} four {rb[i]!r}  (d - 5) % 4 = {(d - 5) % 4}")
        break
else:
    print("sums_close identical")
cnt = {}
for d in range(nn):
    ra, rb = a[d, : nn - d], b[d, : nn - d]
    k = int(np.sum((np.isfinite(ra) | np.isfinite(rb)) & (ra != rb)))
    if k: cnt[d] = k
print("diagonals with differing sums_close (d: cells):", dict(list(cnt.items())[:24]))
