"""Experiment (see gen4_probe.py): T_GEN_D (slots 36-37, {max, sum} per cell) and the X4 planes (slots 10-13) right
after the inside sweep, three against four diagonals per k_tlane_gen launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rna_algos_amd import workloads as W
from rna_algos_amd.utils import FoldScoreSets
from rna_algos_amd.mccaskill_algo import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
s = W.synthetic_seq(n, 13 * n + 5)
P = FoldScoreSets.synthetic(1)
gen, x4 = {}, {}
for gb in (3, 4):
    for what, slot, ns in (("gen", 36, 2), ("x4", 10, 4)):
        ctx = Context(P, device=0)
        ctx.set("summation_mode", 1); ctx.set("tree_lane", 2); ctx.set("tree_gen_batch", gb)
        path = f"/tmp/{what}_{gb}.bin"
        os.environ["RNAMC_DUMP_MID"] = f"{slot},{ns},{path}"
        ctx.bpp_batch([s], False, False)
        raw = np.fromfile(path, dtype=np.uint8)
        nn, ld, msz = (int(x) for x in raw[:24].view(np.uint64))
        f = raw[24:].view(np.float32)
        if what == "gen":
            gen[gb] = f[: 2 * nn * ld].reshape(nn, ld, 2)
        else:
            x4[gb] = [f[c * msz: c * msz + nn * ld].reshape(nn, ld) for c in range(4)]
        ctx.close()
d = 8
ga, gbb = gen[3][d], gen[4][d]
val = lambda g: np.where(g[:, 1] > 0, g[:, 0] + np.log(np.maximum(g[:, 1], 1e-38)), -np.inf)
va, vb = val(ga), val(gbb)
idx = np.where(va != vb)[0][:6]
print("GEN(8) cells differing:", int(np.sum(va != vb)), "first:", [(int(i), float(va[i]), float(vb[i])) for i in idx])
for c in range(4):
    for dd in (3, 4, 5):
        a, b = x4[3][c][dd, : nn - dd], x4[4][c][dd, : nn - dd]
        print(f"X4 plane {c} diagonal {dd}: finite cells {int(np.isfinite(a).sum())} / {int(np.isfinite(b).sum())}, differing {int(np.sum(a != b) - np.sum(np.isnan(a) & np.isnan(b)))}")
