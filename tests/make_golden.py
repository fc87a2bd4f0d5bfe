"""Generates tests/golden/*.json|npz from the CPU oracle with the seeded synthetic
tables ("self-golden, synthetic tables": the reference itself cannot run here and
holds no golden vectors for this path — SURVEY.md §8c).  Usage:
    python tests/make_golden.py [small|container|batch|batch2k|n1024|n4096_turner|n4096_contra|exact|exact4096]
`exact` writes the f64 fixtures of the tree-order mode (oracle/mccaskill_exact.c: the oracle's own
loops with Score = double and an exact logsumexp): ln Z, the probability of every 97th present
pair and the sha256 of the key set, for n = 1024 (both models, seed 1024) and n = 2048 Turner.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from rna_algos_amd.utils import FoldScoreSets, read_fasta  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
PARAM_SEED = 1
EXACT_STRIDE = 97  # every 97th present pair (packed diagonal-major order) goes into a fixture


def digest(packed):
    """sha256 of the f32 bits with the libm-exp branch (p >= 0.9999) canonicalised."""
    a = np.array(packed, dtype=np.float32, copy=True)
    a[a >= 0.9999] = 1.0
    return hashlib.sha256(a.tobytes()).hexdigest()


def summary(packed, logz, n, secs):
    pres = packed >= -0.5
    return {"n": int(n), "log_partition_bits": int(np.float32(logz).view(np.uint32)),
            "log_partition": float(logz), "sha256": digest(packed),
            "present": int(pres.sum()), "max": float(packed.max()),
            "sum_present": float(packed[pres].astype(np.float64).sum()), "oracle_seconds": secs}


def main(which):
    P = FoldScoreSets.synthetic(PARAM_SEED)
    if which == "small":
        recs = read_fasta(os.path.join(GOLD, "sampled_trnas.fa"))
        arrs = {}
        for idx, (_, s) in enumerate(recs):
            for contra in (0, 1):
                out, lz = O.bpp(P.ptr, s, contra, 0)
                arrs[f"trna{idx}_{'contra' if contra else 'turner'}"] = out
                arrs[f"trna{idx}_{'contra' if contra else 'turner'}_logz"] = np.array([lz], np.float32)
        np.savez_compressed(os.path.join(GOLD, "trna_bpp_synthetic_seed1.npz"), **arrs)
        return
    if which == "container":
        # the RNAMCGLD container of bindings/rust/dump_tables.rs, written from the oracle under
        # synthetic tables: 6 tRNAs + a seeded random n = 120, three model variants each
        import golden_io
        recs = read_fasta(os.path.join(GOLD, "sampled_trnas.fa"))
        seqs = [s for _, s in recs] + [O.splitmix_seq(120, 120)]
        out = []
        for s in seqs:
            for contra, short in ((False, False), (True, False), (True, True)):
                out.append((s, contra, short, O.bpp(P.ptr, s, contra, short)[0]))
        golden_io.write(os.path.join(GOLD, "synthetic_goldens.bin"), out)
        return
    if which == "batch":
        # sequences 0, 1, 3 of the 10k-sequence bench batch (lengths 1653, 1024, 531)
        from rna_algos_amd import workloads as W
        lens = W.batch_lengths(8)
        res = {}
        for idx in (0, 1, 3):
            s = W.synthetic_seq(int(lens[idx]), (10000 << 32) + idx)
            for contra in (0, 1):
                t0 = time.time()
                out, lz = O.bpp(P.ptr, s, contra, 0)
                res[f"batch{idx}_{'contra' if contra else 'turner'}"] = summary(out, lz, len(s), time.time() - t0)
        with open(os.path.join(GOLD, "checksums_batch.json"), "w") as fh:
            json.dump({"param_seed": PARAM_SEED, "cases": res}, fh, indent=1)
        print(json.dumps(res, indent=1))
        return
    if which == "batch2k":
        # one of the LONGEST members of the 10k bench batch: it sits inside the first lock-step
        # group (the ~740 longest sequences, where the workspace cap and the multi-kernel
        # outside path engage)
        from rna_algos_amd import workloads as W
        lens = W.batch_lengths(10000)
        idx = int(np.argsort(-lens, kind="stable")[5])
        s = W.synthetic_seq(int(lens[idx]), (10000 << 32) + idx)
        res = {}
        for contra in (0, 1):
            t0 = time.time()
            out, lz = O.bpp(P.ptr, s, contra, 0)
            res[f"batch{idx}_{'contra' if contra else 'turner'}"] = summary(out, lz, len(s), time.time() - t0)
            with open(os.path.join(GOLD, "checksums_batch2k.json"), "w") as fh:
                json.dump({"param_seed": PARAM_SEED, "batch_count": 10000, "cases": res}, fh, indent=1)
        print(json.dumps(res, indent=1))
        return
    if which in ("exact", "exact4096"):
        # (exact4096: BASELINE.json configs[2], n = 4096 Turner — about 40 minutes of one core)
        for n, seed, contra in (((1024, 1024, 1), (1024, 1024, 0), (2048, 2048, 0)) if which == "exact"
                                else ((4096, 4096, 0),)):
            s = O.splitmix_seq(n, seed)
            t0 = time.time()
            xb, xz = O.exact_bpp(P.ptr, s, contra, 0)
            secs = time.time() - t0
            pres = np.flatnonzero(xb >= -0.5)
            pick = pres[::EXACT_STRIDE]
            name = f"exact_n{n}_seed{seed}_{'contra' if contra else 'turner'}"
            np.savez_compressed(os.path.join(GOLD, name + ".npz"),
                                n=np.array([n], np.int64), seed=np.array([seed], np.int64),
                                contra=np.array([contra], np.int64),
                                param_seed=np.array([PARAM_SEED], np.int64),
                                log_partition=np.array([xz], np.float64),
                                present=np.array([pres.size], np.int64),
                                keyset_sha256=np.frombuffer(
                                    hashlib.sha256(pres.astype(np.uint32).tobytes()).digest(), np.uint8),
                                index=pick.astype(np.uint32), prob=xb[pick].astype(np.float64),
                                max_prob=np.array([xb[pres].max()], np.float64),
                                sum_present=np.array([xb[pres].sum()], np.float64),
                                oracle_seconds=np.array([secs], np.float64))
            print(name, "ln Z", xz, "present", pres.size, "picked", pick.size, f"{secs:.1f} s", flush=True)
        return
    cases = {"n1024": [(1024, 1024, 1), (1024, 1024, 0)], "n4096_turner": [(4096, 4096, 0)],
             "n4096_contra": [(4096, 4096, 1)]}[which]
    res = {}
    for n, seed, contra in cases:
        s = O.splitmix_seq(n, seed)
        t0 = time.time()
        out, lz = O.bpp(P.ptr, s, contra, 0)
        res[f"n{n}_seed{seed}_{'contra' if contra else 'turner'}"] = summary(out, lz, n, time.time() - t0)
    path = os.path.join(GOLD, f"checksums_{which}.json")
    with open(path, "w") as fh:
        json.dump({"param_seed": PARAM_SEED, "cases": res}, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "small")
