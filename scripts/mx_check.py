"""Mid-field products on the matrix cores (k_tree_mid_mx, knob tree_mid_mx) against the VALU form
(k_tree_mid) of the tree-order mode: same terms, another grouping and scaling — equal to f32 rounding.
usage: mx_check.py [lengths,comma]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rna_algos_amd import workloads as W  # noqa: E402
from rna_algos_amd.utils import FoldScoreSets  # noqa: E402
from rna_algos_amd.mccaskill_algo import Context  # noqa: E402


def dev(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    ka, kb = a >= -0.5, b >= -0.5
    both = ka & kb
    return bool(np.array_equal(ka, kb)), float(np.abs(a[both] - b[both]).max()) if both.any() else 0.0


def main():
    P = FoldScoreSets.synthetic(1)
    ctx = Context(P, device=0)
    ctx.set("summation_mode", 1)
    lens = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "200,257,300,410,640,1024,1500".split(","))]
    bad = 0
    for contra, short in ((False, False), (True, False), (True, True)):
        seqs = [W.synthetic_seq(n, 13 * n + 5) for n in lens]
        for lane, sync in ((0, 0), (2, 0), (2, 1)):
            ctx.set("tree_lane", lane)
            ctx.set("tree_mid_sync", sync)
            res = {}
            for mx in (0, 1):
                ctx.set("tree_mid_mx", mx)
                res[mx] = ctx.bpp_batch(seqs, contra, short)
            for n, a, b0, za, zb in zip(lens, res[1][0], res[0][0], res[1][1], res[0][1]):
                same, dp = dev(a.packed, b0.packed)
                dz = abs(float(za) - float(zb))
                tol = 2 * (2e-5 + 2e-7 * n)
                flag = "" if (same and dp <= tol and dz <= 3e-6 * max(1.0, abs(float(zb)))) else "   <-- DIFFERS"
                bad += bool(flag)
                print(f"contra={contra} short={short} lane={lane} sync={sync} n={n}: keys {same} max|dp| {dp:.3e} "
                      f"|dlnZ| {dz:.3e} (lnZ {float(zb):.4f}){flag}", flush=True)
    print("BAD" if bad else "ALL OK", bad)
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
