"""Band launches of the tree-order sweep (rnamc_tree_band.h, knob tree_steps) against the
two-diagonals-per-launch sweep of the same mode: same terms, another grouping — equal to f32
rounding; then the n = 4096 timing of both."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
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
    lens = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "200,257,300,410,640,1024".split(","))]
    bad = 0
    for contra, short in ((False, False), (True, False), (True, True)):
        seqs = [W.synthetic_seq(n, 13 * n + 5) for n in lens]
        for group in (False, True):
            res = {}
            for steps in (0, 1):
                ctx.set("tree_steps", steps)
                if group:
                    res[steps] = ctx.bpp_batch(seqs, contra, short)
                else:
                    mats, zs = [], []
                    for s in seqs:
                        m, z = ctx.bpp_batch([s], contra, short)
                        mats.append(m[0])
                        zs.append(z[0])
                    res[steps] = (mats, zs)
            for n, a, b0, za, zb in zip(lens, res[1][0], res[0][0], res[1][1], res[0][1]):
                same, dp = dev(a.packed, b0.packed)
                dz = abs(float(za) - float(zb))
                tol = 2 * (2e-5 + 2e-7 * n)
                flag = "" if (same and dp <= tol and dz <= 3e-6 * max(1.0, abs(float(zb)))) else "   <-- DIFFERS"
                bad += bool(flag)
                print(f"contra={contra} short={short} group={group} n={n}: keys {same} max|dp| {dp:.3e} "
                      f"|dlnZ| {dz:.3e} (lnZ {float(zb):.4f}){flag}", flush=True)
    print("BAD" if bad else "ALL OK", bad)
    if os.environ.get("TIME4096", "1") == "1" and not bad:
        n = 4096
        s = W.synthetic_seq(n, n)
        d = torch.device("cuda:0")
        b = torch.from_numpy(np.ascontiguousarray(s)).to(d)
        o = torch.empty(n * (n + 1) // 2, dtype=torch.float32, device=d)
        z = torch.empty(1, dtype=torch.float32, device=d)
        off = np.array([0, n], dtype=np.uint64)
        oo = np.array([0, n * (n + 1) // 2], dtype=np.uint64)
        ctx.set("profile", 1)
        outs = {}
        for contra in (False, True):
            for steps in (0, 1):
                ctx.set("tree_steps", steps)
                ms = []
                for r in range(5):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    ctx.bpp_batch_device(1, b.data_ptr(), off, contra, False, o.data_ptr(), oo, z.data_ptr(), 0)
                    torch.cuda.synchronize()
                    ms.append((time.perf_counter() - t0) * 1e3)
                st = ctx.stats()
                outs[(contra, steps)] = (o.cpu().numpy().copy(), float(z[0]))
                print(f"n=4096 contra={contra} tree_steps={steps}: {np.median(ms[1:]):.2f} ms (calls {[round(x, 1) for x in ms]}) "
                      f"inside {st['ms_inside']:.2f} outside {st['ms_outside']:.2f} launches "
                      f"{st['launches_inside']}+{st['launches_outside']}", flush=True)
            same, dp = dev(outs[(contra, 1)][0], outs[(contra, 0)][0])
            print(f"   n=4096 contra={contra}: band vs pair launches keys {same} max|dp| {dp:.3e} "
                  f"dlnZ {outs[(contra, 1)][1] - outs[(contra, 0)][1]:.3e}")
    ctx.close()


if __name__ == "__main__":
    main()
