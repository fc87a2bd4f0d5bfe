"""Ad-hoc timing probe (not the contract bench): a few workloads through the
device-resident API with event-timed inside/outside sweeps."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rna_algos_amd import workloads as WL0  # noqa: E402
from rna_algos_amd.utils import FoldScoreSets  # noqa: E402
from rna_algos_amd.mccaskill_algo import Context  # noqa: E402


def batch_lengths(count, master_seed=10000):
    mask = (1 << 64) - 1
    state = master_seed
    lens = []
    for _ in range(count):
        state = (state + 0x9E3779B97F4A7C15) & mask
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z = z ^ (z >> 31)
        lens.append(256 + z % 1793)
    return lens


def run(ctx, seqs, contra, reps=2, label=""):
    dev = torch.device("cuda:0")
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    out_offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens * (lens + 1) // 2, out=out_offsets[1:])
    bases = torch.from_numpy(np.concatenate(seqs)).to(dev)
    out = torch.empty(int(out_offsets[-1]), dtype=torch.float32, device=dev)
    logz = torch.empty(len(seqs), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.time()
        ctx.bpp_batch_device(len(seqs), bases.data_ptr(), offsets, contra, False, out.data_ptr(),
                             out_offsets, logz.data_ptr(), st)
        torch.cuda.synchronize()
        dt = time.time() - t0
        s = ctx.stats()
        T = float(sum(int(n) * (int(n) ** 2 - 1) / 6 for n in lens))
        print(f"{label} contra={contra} rep{r}: {dt*1e3:.1f} ms  nt/s={lens.sum()/dt:.3e}  "
              f"inside={s['ms_inside']:.1f} outside={s['ms_outside']:.1f} other={s['ms_other']:.1f} ms "
              f"groups={s['n_groups']} T={T:.3e} ns/T={dt*1e9/T:.3f}", flush=True)
        if s["launches_outside_main"]:
            print(f"   per kernel: main {s['ms_outside_main']:.1f} ms / {s['launches_outside_main']}, "
                  f"tail {s['ms_outside_tail']:.1f} / {s['launches_outside_tail']}, "
                  f"small {s['ms_outside_small']:.1f} / {s['launches_outside_small']}", flush=True)
    return out, logz


def main():
    P = FoldScoreSets.synthetic(1)
    ctx = Context(P, device=0)
    ctx.set("profile", 1)
    if os.environ.get("ROLES"):
        ctx.set("debug_roles", int(os.environ["ROLES"]))
    if os.environ.get("WSGB"):
        ctx.set("group_ws_bytes", int(os.environ["WSGB"]) << 30)
    if os.environ.get("ORDER_IN"):
        ctx.set("order_inside", int(os.environ["ORDER_IN"]))
    if os.environ.get("ORDER_OUT"):
        ctx.set("order_outside", int(os.environ["ORDER_OUT"]))
    if os.environ.get("DUAL"):
        ctx.set("dual_outside", int(os.environ["DUAL"]))
    if os.environ.get("DUALDIAG"):
        ctx.set("dual_max_diag", int(os.environ["DUALDIAG"]))
    if os.environ.get("DUALMIN"):
        ctx.set("dual_min_cells", int(os.environ["DUALMIN"]))
    if os.environ.get("FUSE"):
        ctx.set("fuse_inside", int(os.environ["FUSE"]))
    for kv in filter(None, os.environ.get("SETS", "").split(",")):
        k, v = kv.split("=")
        ctx.set(k, int(v))
    if os.environ.get("BLOCK"):
        ctx.set("block_threads", int(os.environ["BLOCK"]))
    what = sys.argv[1:] or ["n1024", "n4096", "batch256"]
    for w in what:
        if w == "n1024":
            run(ctx, [WL0.synthetic_seq(1024, 1024)], True, label="n1024")
            run(ctx, [WL0.synthetic_seq(1024, 1024)], False, label="n1024")
        elif w == "n4096":
            run(ctx, [WL0.synthetic_seq(4096, 4096)], False, label="n4096")
            run(ctx, [WL0.synthetic_seq(4096, 4096)], True, reps=1, label="n4096")
        elif w.startswith("multi"):
            # multi<cnt>x<n>: cnt sequences of n nt in one group (latency-form crossover)
            cnt, n = (int(x) for x in w[5:].split("x"))
            seqs = [WL0.synthetic_seq(n, 77 * n + s) for s in range(cnt)]
            run(ctx, seqs, False, label=w)
            run(ctx, seqs, True, label=w)
        elif w.startswith("top"):
            # the `cnt` longest sequences of the 10k batch: what one lock-step group really holds
            cnt = int(w[3:])
            from rna_algos_amd import workloads as WL
            lens = WL.batch_lengths(10000)
            order = np.argsort(-lens, kind="stable")[:cnt]
            seqs = [WL.synthetic_seq(int(lens[s]), (10000 << 32) + int(s)) for s in order]
            ctx.set("group_max_seqs", cnt)
            run(ctx, seqs, False, reps=1, label=w)
        elif w.startswith("bot"):
            # the `cnt` shortest sequences of the 10k batch (2-loop work dominates there)
            cnt = int(w[3:])
            from rna_algos_amd import workloads as WL
            lens = WL.batch_lengths(10000)
            order = np.argsort(lens, kind="stable")[:cnt]
            seqs = [WL.synthetic_seq(int(lens[s]), (10000 << 32) + int(s)) for s in order]
            print("lengths", int(lens[order].min()), int(lens[order].max()))
            run(ctx, seqs, False, reps=2, label=w)
        elif w.startswith("batch"):
            cnt = int(w[5:])
            lens = batch_lengths(10000)[:cnt]
            rng = np.random.default_rng(5)
            seqs = [rng.integers(0, 4, n).astype(np.uint8) for n in lens]
            gms = [int(x) for x in os.environ.get("GSIZES", "256,1024").split(",")]
            for gm in gms:
                ctx.set("group_max_seqs", gm)
                run(ctx, seqs, False, reps=1, label=f"{w} group{gm}")
            if os.environ.get("CONTRA", "1") == "1":
                run(ctx, seqs, True, reps=1, label=f"{w} group{gms[-1]}")


if __name__ == "__main__":
    main()
