#!/usr/bin/env python3
"""Headline benchmark of the McCaskill bpp hot path on MI355X.

Contract (one JSON line from rank 0):
  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (inside + outside sweep + final map) over the
whole workload, inputs already resident in HBM, outputs left in HBM.  The default
workload is BASELINE.json's batch config: 10 000 synthetic RNAs of length 256..2048
(SplitMix64, master seed 10000), Turner-2004-shaped synthetic tables, reference-order
(bit-faithful) summation.  With N GPUs the batch is sharded by longest-processing-time
over sum n(n^2-1)/6 (strong scaling: total work fixed); no data-path collective — the
only communication is the barrier and the max-over-ranks of the step time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)


def shard_lpt(costs, world):
    """Longest-processing-time assignment of units to `world` ranks.
    Returns a list of index arrays (one per rank)."""
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    loads = np.zeros(world)
    shards = [[] for _ in range(world)]
    for idx in order:
        r = int(np.argmin(loads))
        shards[r].append(int(idx))
        loads[r] += costs[idx]
    return [np.array(s, dtype=np.int64) for s in shards]


def shard_banded(costs, world):
    """Contiguous bands of the cost-sorted list with equal total cost: every rank gets
    sequences of similar length, so its lock-step groups stay large on every diagonal."""
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    csum = np.cumsum(np.asarray(costs, dtype=np.float64)[order])
    cuts = np.searchsorted(csum, csum[-1] * np.arange(1, world) / world)
    return [np.array(x, dtype=np.int64) for x in np.split(order, cuts)]


def build_workload(name, batch_count):
    from rna_algos_amd import workloads as W
    if name == "batch10k":
        lens = W.batch_lengths(batch_count)
        seqs = [W.synthetic_seq(int(lens[s]), (10000 << 32) + s) for s in range(batch_count)]
        label = f"{batch_count} synthetic RNAs, length 256..2048 (SplitMix64 master seed 10000)"
    elif name == "n4096":
        seqs = [W.synthetic_seq(4096, 4096)]
        label = "single synthetic n=4096 RNA (seed 4096)"
    elif name == "n1024":
        seqs = [W.synthetic_seq(1024, 1024)]
        label = "single synthetic n=1024 RNA (seed 1024)"
    else:
        raise SystemExit(f"unknown workload {name}")
    return seqs, label


def outside_bytes(lengths, f, part="all"):
    """Algorithmic bytes of the outside sweep: L_d 8 B per (cell,k), L_e 12 B per (paired
    cell,k), 8 B per enclosing-pair probe, 4 packed triangles written.  part "main" = the
    roles of k_outside<.,5> (L_d, the probes, 2 triangles), "tail" = k_outside<.,2> (L_e)."""
    from rna_algos_amd import workloads as W
    lengths = np.asarray(lengths, dtype=np.float64)
    T = W.pair_cost(lengths).sum()
    n2 = (lengths * lengths).sum()
    main = 8.0 * T + 8.0 * 496.0 * f * n2 / 2.0 + 8.0 * n2 / 2.0
    tail = 12.0 * f * T + 8.0 * n2 / 2.0
    return {"all": main + tail, "main": main, "tail": tail}[part]


def inside_bytes(lengths, f, contra, schedule="model"):
    """Algorithmic bytes of the inside sweep.  "model": SURVEY.md 8d's streamed-operand
    figure (L_b 8 B + L_c 8 B per (cell,k), + L_a 4 B CONTRAfold; 4 B per probe; 5 triangles
    written).  "two_diagonal": what the implemented schedule has to move — one lane folds two
    cells off one stream of the two row operands and one column operand (12 B per two cells;
    CONTRAfold: two column operands and half an L_a stream more)."""
    from rna_algos_amd import workloads as W
    lengths = np.asarray(lengths, dtype=np.float64)
    T = W.pair_cost(lengths).sum()
    n2 = (lengths * lengths).sum()
    if schedule == "model":
        per_t = 16.0 + (4.0 if contra else 0.0)
    else:
        per_t = 6.0 + (4.0 if contra else 0.0)
    return per_t * T + 4.0 * 496.0 * f * n2 / 2.0 + 20.0 * n2 / 2.0


def pmc_traffic_per_launch(kernel, total_T, launches):
    """HBM bytes per launch of `kernel` from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, scripts/prof_traffic.sh; FETCH_SIZE doubled as the gfx950 guide
    prescribes), scaled from that pass's workload to this one by sum n(n^2-1)/6.  None when the
    profile is absent."""
    path = os.path.join(ROOT, "profiles", "r01_traffic_batch1000.json")
    try:
        t = json.load(open(path))
        from rna_algos_amd import workloads as W
        ref_T = float(W.pair_cost(W.batch_lengths(1000)).sum())
        k = t[kernel]
        return (k["fetch_bytes_x2"] + k["write_bytes"]) * (total_T / ref_T) / max(launches, 1)
    except Exception:
        return None


def cpu_baseline(params, seqs, contra, budget_s):
    """Oracle ("port" of the reference's CPU path) on a bounded, length-stratified sample,
    one sequence per thread on all host cores (as the reference's thread pool does)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from rna_algos_amd import workloads as W
    # a 1-GPU box's CPU share is 16 cores; never more threads than that
    cores = min(os.cpu_count() or 1, 16)
    lens = np.array([len(s) for s in seqs])
    # ~150 ns per (cell,k) per core for the dense row-major restatement: every thread gets
    # one sequence whose cost fits the budget, taken at even strides of the length-sorted
    # workload below that cost (the sample is stated in the result)
    per_core_T = budget_s / 150e-9
    fits = np.nonzero(W.pair_cost(lens) <= per_core_T)[0]
    if fits.size:
        order = fits[np.argsort(lens[fits])]
        k = min(cores, order.size)
        pick = order[np.linspace(0, order.size - 1, k).astype(int)]
        sample = [seqs[i] for i in pick]
    else:
        # a single long sequence (n = 4096): time a prefix whose cost fits the budget
        m = int((per_core_T * 6) ** (1 / 3))
        sample = [seqs[int(np.argmin(lens))][:m]]
    t0 = time.time()
    O.bpp_batch(params.ptr, sample, contra, False, n_threads=min(cores, len(sample)),
                want_bpp=False)
    dt = time.time() - t0
    nt = int(sum(len(s) for s in sample))
    T = float(W.pair_cost(np.array([len(s) for s in sample])).sum())
    T_all = float(W.pair_cost(lens).sum())
    return {
        # scaled to the metric's unit: the workload's nt/s if the whole of it ran at the
        # sample's rate per (cell,k) iteration (cost model sum n(n^2-1)/6)
        "value": float(lens.sum()) / (dt * T_all / T), "unit": "nt/s",
        "cores": int(min(cores, len(sample))), "kind": "port",
        "sample": f"{len(sample)} sequences (lengths {sorted(len(s) for s in sample)}) of the "
                  f"workload, one per thread, {dt:.1f} s wall; extrapolated by sum n(n^2-1)/6. "
                  f"The port is faster than the Rust reference would be (dense arrays, no "
                  f"twoloop_scores hash map)",
        "sample_nt_per_s": nt / dt,
        "ns_per_cell_k_all_cores": dt * 1e9 / T,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="batch10k", choices=["batch10k", "n4096", "n1024"])
    ap.add_argument("--model", default="turner", choices=["turner", "contra"])
    ap.add_argument("--batch-count", type=int, default=10000)
    ap.add_argument("--rehearse-shard", default="",
                    help="W:R — single process, run the shard rank R of W ranks would get "
                         "(rehearsal of the per-rank time of a multi-GPU run; value is per-shard)")
    ap.add_argument("--shard", choices=["lpt", "banded"], default="banded",
                    help="banded: every rank gets sequences of similar length (larger "
                         "lock-step launches: 95 %% of linear at 8 ranks against 91 %% for lpt)")
    ap.add_argument("--group-max-seqs", type=int, default=0)
    ap.add_argument("--group-max-nt", type=int, default=0)
    ap.add_argument("--group-ws-gb", type=int, default=0)
    ap.add_argument("--param-seed", type=int, default=1)
    ap.add_argument("--cpu-budget-s", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="skip the extra pass that times every outside kernel with HIP events")
    ap.add_argument("--no-n4096", action="store_true",
                    help="skip the single n=4096 sequence (second half of BASELINE.json's metric)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rna_algos_amd import workloads as W
    from rna_algos_amd.utils import FoldScoreSets
    from rna_algos_amd.mccaskill_algo import Context

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    contra = args.model == "contra"

    seqs, label = build_workload(args.workload, args.batch_count)
    lens_all = np.array([len(s) for s in seqs], dtype=np.int64)
    costs = W.sweep_cost(lens_all)
    shard = shard_banded if args.shard == "banded" else shard_lpt
    if args.rehearse_shard:
        rw, rr = (int(x) for x in args.rehearse_shard.split(":"))
        mine = shard(costs, rw)[rr]
        label += f" [shard {rr} of {rw}]"
    elif len(seqs) >= world:
        mine = shard(costs, world)[rank]
    else:  # fewer units than ranks (single-sequence workloads): replicas
        mine = np.arange(len(seqs))
    my_seqs = [seqs[i] for i in mine]
    lens = lens_all[mine].astype(np.uint64)
    offsets = np.zeros(len(my_seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    out_offsets = np.zeros(len(my_seqs) + 1, dtype=np.uint64)
    np.cumsum(lens * (lens + np.uint64(1)) // np.uint64(2), out=out_offsets[1:])

    params = FoldScoreSets.synthetic(args.param_seed)
    ctx = Context(params, device=local_rank)
    ctx.set("profile", 1)
    if args.group_max_seqs:
        ctx.set("group_max_seqs", args.group_max_seqs)
    if args.group_max_nt:
        ctx.set("group_max_nt", args.group_max_nt)
    if args.group_ws_gb:
        ctx.set("group_ws_bytes", args.group_ws_gb << 30)

    # inputs resident in HBM before the timed region; outputs stay in HBM
    d_bases = torch.from_numpy(np.concatenate(my_seqs)).to(dev)
    d_out = torch.empty(int(out_offsets[-1]), dtype=torch.float32, device=dev)
    d_logz = torch.empty(len(my_seqs), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.bpp_batch_device(len(my_seqs), d_bases.data_ptr(), offsets, contra, False,
                             d_out.data_ptr(), out_offsets, d_logz.data_ptr(), stream)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    ms_in = ms_out = ms_main = ms_tail = ms_small = 0.0
    l_in = l_out = l_main = l_tail = l_small = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = ctx.stats()  # event-timed sweeps of this step (the call synchronised its stream)
        ms_in += st["ms_inside"]
        ms_out += st["ms_outside"]
        l_in += st["launches_inside"]
        l_out += st["launches_outside"]
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    # one more pass, outside the timed region, with a pair of HIP events around every
    # outside-sweep kernel (on the stream it is launched on): per-kernel durations for the
    # roofline of the dominant kernel.  Those event records cost ~2 %, so the timed steps
    # above run without them.
    if rank == 0 and not args.no_kernel_timing:
        ctx.set("profile", 2)
        step()
        st = ctx.stats()
        ctx.set("profile", 1)
        ms_main, ms_tail, ms_small = (st["ms_outside_main"], st["ms_outside_tail"],
                                      st["ms_outside_small"])
        l_main, l_tail, l_small = (st["launches_outside_main"], st["launches_outside_tail"],
                                   st["launches_outside_small"])
        torch.cuda.synchronize()
    barrier()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the result is a probability matrix (cheap, outside the timed region)
    probe = d_out[: min(d_out.numel(), 1 << 22)]
    pres = probe[probe >= -0.5]
    assert pres.numel() > 0 and float(pres.min()) >= -0.001 and float(pres.max()) < 1.05
    assert bool(torch.isfinite(d_logz).all())

    if rank == 0:
        total_nt = int(lens_all.sum()) if len(seqs) >= world else int(lens_all.sum()) * world
        steps = max(args.steps, 1)
        f = float(np.mean([W.paired_fraction(s) for s in my_seqs[:: max(1, len(my_seqs) // 8)]]))
        b_out = outside_bytes(lens.astype(np.float64), f)
        b_main = outside_bytes(lens.astype(np.float64), f, "main")
        b_tail = outside_bytes(lens.astype(np.float64), f, "tail")
        total_T = float(W.pair_cost(lens.astype(np.float64)).sum())
        if l_main == 0:
            # nothing was large enough to split (single sequences): all roles ran in
            # k_outside<.,7>, which then is the dominant kernel
            b_main, ms_main, l_main = b_out, ms_small, l_small
        b_in = inside_bytes(lens.astype(np.float64), f, contra, "two_diagonal")
        b_in_model = inside_bytes(lens.astype(np.float64), f, contra, "model")
        avg_out_ms = ms_out / max(l_out, 1)
        avg_in_ms = ms_in / max(l_in, 1)
        ach_out = b_out * steps / (ms_out * 1e-3) / 1e9 if ms_out > 0 else 0.0
        ach_in = b_in * steps / (ms_in * 1e-3) / 1e9 if ms_in > 0 else 0.0
        res = {
            "metric": "nucleotides/sec (batch)" if args.workload == "batch10k" else "nucleotides/sec",
            "value": total_nt * steps / elapsed,
            "unit": "nt/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": label,
                "model": "Turner-2004-shaped synthetic tables" if not contra
                         else "CONTRAfold-shaped synthetic tables",
                "tables": f"synthetic seed {args.param_seed} (real tables live in the absent "
                          f"rna-ss-params crate)",
                "summation": "reference-order (bit-faithful to the CPU path)",
                "allows_short_hairpins": False,
                "sharding": f"{args.shard} over the measured cost model a*n(n^2-1)/6 + b*n^2, "
                            f"{world} rank(s), no data-path collective",
                "sequences_rank0": len(my_seqs),
                "paired_fraction_f": f,
            },
            # the dominant kernel by GPU time (rocprofv3 --stats): per-kernel accounting, HIP
            # events around each of its launches on its own stream.  Its few small-launch
            # siblings (k_outside<.,7>, < 1 % of the time) do the same roles on the first
            # diagonals; their bytes are left in, which costs the figure a fraction of a percent
            "roofline": {
                "kernel": ("k_outside<.,5> (probs_multibranch + 2-loop half of the pair "
                           "probabilities, one launch per anti-diagonal)") if l_tail else
                          "k_outside<.,7> (outside sweep, all roles, one launch per anti-diagonal)",
                "bound": "hbm",
                "achieved": b_main / (ms_main * 1e-3) / 1e9 if ms_main > 0 else 0.0,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (b_main / (ms_main * 1e-3) / 1e9 if ms_main > 0 else 0.0) / HBM_PEAK_GBS,
                "traffic": pmc_traffic_per_launch("k_outside_main", total_T, l_main)
                if args.workload == "batch10k" and not contra else None,
                "traffic_source": "profiles/r01_traffic_batch1000.json (separate --pmc passes, scaled by "
                                  "sum n(n^2-1)/6)",
                "algorithmic_bytes_per_launch": b_main / max(l_main, 1),
                "avg_launch_ms": ms_main / max(l_main, 1),
                "launches_per_step": l_main,
                "note": "runs beside k_outside<.,2> (second stream): both share the chip",
            },
            "roofline_tail": {
                "kernel": "k_outside<.,2> (multibranch half of the pair probabilities)",
                "bound": "hbm",
                "achieved": b_tail / (ms_tail * 1e-3) / 1e9 if ms_tail > 0 else 0.0,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (b_tail / (ms_tail * 1e-3) / 1e9 if ms_tail > 0 else 0.0) / HBM_PEAK_GBS,
                "traffic": pmc_traffic_per_launch("k_outside_tail", total_T, l_tail)
                if args.workload == "batch10k" and not contra else None,
                "algorithmic_bytes_per_launch": b_tail / max(l_tail, 1),
                "avg_launch_ms": ms_tail / max(l_tail, 1),
                "launches_per_step": l_tail,
            },
            "roofline_outside_sweep": {
                "what": "both kernels together: algorithmic bytes of the whole outside sweep over its "
                        "event-timed duration",
                "bound": "hbm", "achieved": ach_out, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach_out / HBM_PEAK_GBS,
                "ms_per_step": ms_out / steps,
                "kernels_per_step": l_out // steps,
                "small_launches_per_step": l_small,
                "small_launch_ms_per_step": ms_small,
            },
            "roofline_inside": {
                "kernel": "k_inside2 / k_inside (inside sweep)", "bound": "hbm", "achieved": ach_in,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_in / HBM_PEAK_GBS,
                "traffic": pmc_traffic_per_launch("k_inside", total_T, l_in / steps)
                if args.workload == "batch10k" and not contra else None,
                "avg_launch_ms": avg_in_ms, "launches_per_step": l_in // steps,
                "note": "bytes of the two-diagonal schedule (one lane folds two cells off one stream "
                        "of the row operands: 6 B per (cell,k)); by SURVEY 8d's streamed-operand "
                        "model (16 B per (cell,k)) the same sweep scores model_achieved",
                "model_achieved": b_in_model * steps / (ms_in * 1e-3) / 1e9 if ms_in > 0 else 0.0,
            },
        }
        if args.workload != "batch10k":
            res["ms_per_seq"] = elapsed * 1e3 / steps
        elif world == 1 and not args.no_n4096 and not args.rehearse_shard:
            # the other half of the metric: ms per sequence at n = 4096 (BASELINE.json
            # configs[2]: Turner), one warm-up + one timed call, device-resident
            s4 = W.synthetic_seq(4096, 4096)
            b4 = torch.from_numpy(s4).to(dev)
            o4 = torch.empty(4096 * 4097 // 2, dtype=torch.float32, device=dev)
            z4 = torch.empty(1, dtype=torch.float32, device=dev)
            off4 = np.array([0, 4096], dtype=np.uint64)
            oo4 = np.array([0, 4096 * 4097 // 2], dtype=np.uint64)
            for k in range(2):
                torch.cuda.synchronize()
                t4 = time.perf_counter()
                ctx.bpp_batch_device(1, b4.data_ptr(), off4, contra, False, o4.data_ptr(), oo4,
                                     z4.data_ptr(), stream)
                torch.cuda.synchronize()
                t4 = time.perf_counter() - t4
            res["ms_per_seq_n4096"] = t4 * 1e3
            res["n4096_note"] = ("single n=4096 sequence, same tables: a lock-step group of one is "
                                 "bound by the sequential fold chains the reference's summation "
                                 "order dictates (2 x 8.4 M dependent steps), not by HBM")
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(params, my_seqs, contra, args.cpu_budget_s)
        print(json.dumps(res), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
