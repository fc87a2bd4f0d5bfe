"""Randomised soak of the HIP path against the CPU oracle: ragged batches, all three model
variants, random scheduling knobs.  Test infrastructure, not collected by pytest (minutes
of oracle time); run on the GPU box:  python tests/soak.py [rounds] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402
from rna_algos_amd.utils import FoldScoreSets  # noqa: E402
from rna_algos_amd.mccaskill_algo import Context  # noqa: E402


def same(gpu, ref):
    absent_g, absent_r = gpu < -0.5, ref < -0.5
    if not np.array_equal(absent_g, absent_r):
        return False
    big = ref >= 0.9999  # libm-exp branch of the reference: 1 ulp allowed
    a, b = gpu[~absent_r & ~big], ref[~absent_r & ~big]
    return np.array_equal(a.view(np.uint32), b.view(np.uint32)) and \
        np.all(np.abs(gpu[big] - ref[big]) <= 2e-7)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    bad = 0
    for r in range(rounds):
        P = FoldScoreSets.synthetic(int(rng.integers(1, 50)))
        ctx = Context(P, device=0)
        knobs = {
            "fuse_inside": int(rng.integers(0, 2)),
            "dual_outside": int(rng.integers(0, 2)),
            "dual_min_cells": int(rng.choice([0, 4096, 65536, 262144])),
            "block_threads": int(rng.choice([64, 128, 192, 256])),
            "group_max_seqs": int(rng.choice([3, 17, 8192])),
            "order_outside": int(rng.integers(0, 5)),
            "order_inside": int(rng.integers(0, 3)),
            # round 2: latency forms and the optional 2-loop kernels
            "latency_mode": int(rng.choice([0, 1, 1, 2])),
            "lat_pairs": int(rng.integers(0, 2)),
            "lat_merge": int(rng.integers(0, 2)),
            "lat_zr_ahead": int(rng.integers(0, 2)),
            "lat_inside": int(rng.choice([0, 2])),
            "lat_e_waves": int(rng.choice([0, 64, 2048, 1 << 20])),
        }
        for k, v in knobs.items():
            ctx.set(k, v)
        kind = int(rng.integers(0, 3))
        if kind == 0:    # many short: the two-diagonal / two-kernel forms engage
            lens = rng.integers(1, 200, int(rng.integers(50, 700)))
        elif kind == 1:  # few long
            lens = rng.integers(200, 900, int(rng.integers(2, 12)))
        else:            # low-complexity and tiny
            lens = rng.integers(1, 40, 30)
        alphabet = int(rng.choice([2, 3, 4, 4, 4]))
        seqs = [rng.integers(0, alphabet, int(n)).astype(np.uint8) for n in lens]
        contra, short = [(False, False), (True, False), (True, True)][int(rng.integers(0, 3))]
        if int(rng.integers(0, 4)) == 0:
            # the tables follow the caller's set: a context built on other tables, re-pointed
            ctx.close()
            ctx = Context(FoldScoreSets.synthetic(99), device=0)
            for k, v in knobs.items():
                ctx.set(k, v)
            ctx.sync_params(P)
        mats, logz = ctx.bpp_batch(seqs, contra, short)
        ref, rz = O.bpp_batch(P.ptr, seqs, contra, short, n_threads=16)
        ok = all(same(np.asarray(m.packed), x) for m, x in zip(mats, ref)) and \
            np.array_equal(np.asarray(logz).view(np.uint32), np.asarray(rz).view(np.uint32))
        bad += not ok
        print(f"round {r}: {'ok ' if ok else 'BAD'} seqs={len(seqs)} max_n={int(lens.max())} "
              f"contra={contra} short={short} {knobs}", flush=True)
        ctx.close()
    print("soak:", "all ok" if bad == 0 else f"{bad} BAD rounds")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
