#!/usr/bin/env python3
"""Headline benchmark of the McCaskill bpp hot path on MI355X.

Contract (one JSON line from rank 0):
  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(`--gpus N` with N > 1 outside torchrun starts that launcher itself, as a fresh child
process before anything touches the GPU, and relays its line.)

A "step" is one pass of the hot path (inside + outside sweep + final map) over the
whole workload, inputs already resident in HBM, outputs left in HBM.  The default
workload is BASELINE.json's batch config: 10 000 synthetic RNAs of length 256..2048
(SplitMix64, master seed 10000), Turner-2004-shaped synthetic tables, reference-order
(bit-faithful) summation.  With N GPUs the batch is sharded in cost-balanced bands
(strong scaling: total work fixed); no data-path collective — the only communication is
the barrier and the max-over-ranks of the step time.

Beside `value` (device-resident, as the contract asks) the line carries
`value_with_transfers`: one pass through the host-buffer entry `rnamc_bpp_batch`
(H2D + kernels + D2H — the unit SURVEY.md 8d defines), taken as the first warm-up pass.
The per-kernel HIP-event timing behind `roofline` is taken in the LAST warm-up pass; after
the timed steps the batch members that have committed oracle checksums
(tests/golden/checksums_batch.json) are read back and compared (`parity_check`).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
T_START = time.time()


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
    if len(order) >= world:  # no empty band while there are units to give (rnamc_shard_plan)
        for k in range(world - 1):
            lo = (cuts[k - 1] if k else 0) + 1
            cuts[k] = min(max(cuts[k], lo), len(order) - (world - 1 - k))
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
    cell,k), 8 B per enclosing-pair probe, 4 packed triangles written.  Parts by role:
    "mb" probs_multibranch (L_d), "head" the 2-loop half of the pair probabilities,
    "tail" their multibranch half (L_e); "main" = mb + head (one kernel when not split)."""
    from rna_algos_amd import workloads as W
    lengths = np.asarray(lengths, dtype=np.float64)
    T = W.pair_cost(lengths).sum()
    n2 = (lengths * lengths).sum()
    mb = 8.0 * T + 4.0 * n2 / 2.0
    head = 8.0 * 496.0 * f * n2 / 2.0 + 4.0 * n2 / 2.0
    tail = 12.0 * f * T + 8.0 * n2 / 2.0
    return {"all": mb + head + tail, "main": mb + head, "mb": mb, "head": head, "tail": tail}[part]


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


def pmc_traffic_per_launch(kernel, total_T, launches, contra=False):
    """HBM bytes per launch of `kernel` from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, scripts/prof_traffic.sh; FETCH_SIZE doubled as the gfx950 guide
    prescribes), scaled from that pass's workload to this one by sum n(n^2-1)/6.  None when the
    profile is absent.  The newest round's file wins."""
    names = (("r03_traffic_batch1000_contra.json",) if contra else
             ("r04_traffic_batch1000.json", "r03_traffic_batch1000.json", "r02_traffic_batch1000.json",
              "r01_traffic_batch1000.json"))
    for name in names:
        path = os.path.join(ROOT, "profiles", name)
        try:
            t = json.load(open(path))
            from rna_algos_amd import workloads as W
            ref_T = float(W.pair_cost(W.batch_lengths(1000)).sum())
            k = t[kernel]
            return ((k["fetch_bytes_x2"] + k["write_bytes"]) * (total_T / ref_T) / max(launches, 1),
                    "profiles/" + name)
        except Exception:
            continue
    return None, None


VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9  # 256 CUs x 4 SIMD-32 x 2.4 GHz: f32 VALU issue slots / s


def tree_bytes(n, f, contra, band=64):
    """Bytes of one n-nt sequence in tree-order mode, two ways: SURVEY 8d's streamed-operand
    model of the REFERENCE's loops (the contract's figure) and what the tree-order kernels'
    loads name (banded sweep, two anti-diagonals per launch)."""
    from rna_algos_amd import workloads as W
    T = float(W.pair_cost(n))
    n2h = n * n / 2.0
    b_8d = float(W.algorithmic_bytes([n], contra, f))
    either = 1.0 - (1.0 - f) ** 2  # a launch's cell pair runs L_e when either cell is a pair
    # cubic products: the mid-field goes through k_tree_mid's LDS tiles (47 operand rows of 16 k
    # staged per 256 cells x 16 k: 0.73 B per term); the launches walk the edge, on average
    # 3 band widths of terms per product and cell pair (3 loads per 2 terms inside and for L_e,
    # 4 streams per cell pair for probs_multibranch incl. the neighbour's); one 4-byte gather per
    # generic 2-loop (the plane of its class; every cell's far part is formed once, a launch
    # ahead); dense stores; the two sums_external vectors walk n^2 / 2 terms each (k_tree_ext)
    staged = 47.0 * 16 * 4 / (256 * 16)
    edge = 3.0 * band * n2h / 2.0  # terms walked per product by the launches (per cell PAIR: n2h / 2 pairs)
    inside = staged * T + 12.0 * edge + 4.0 * 496.0 * f * n2h + 15 * 4.0 * n2h + 2 * 4.0 * n2h
    outside = staged * T * (1.0 + either) + (16.0 + 12.0 * either) * edge + 4.0 * 496.0 * f * n2h + 13 * 4.0 * n2h
    # issue slots: ~10 per product term (add, max, fma, exp2 at quarter rate ...), ~45 per 2-loop probe
    valu = 10.0 * T * (1.0 + 1.0 + either) + 45.0 * 496.0 * f * n2h * 2.0
    return {"b_8d": b_8d, "inside": inside, "outside": outside, "valu_slots": valu}


def tree_batch_traffic():
    """PMC record of the tree-order BATCH form (scripts/prof_traffic_tree_batch.sh: FETCH_SIZE x 2 +
    WRITE_SIZE over one pass of a 1 000-sequence slice of the bench batch, and the slice's pass time
    outside the profiler) -> (bytes per pass, seconds per pass, nt, source) or None."""
    path = os.path.join(ROOT, "profiles", "r04_tree_batch_traffic.json")
    try:
        j = json.load(open(path))
        return float(j["total_bytes_x2"]), float(j["pass_ms"]) * 1e-3, int(j["nt"]), "profiles/r04_tree_batch_traffic.json"
    except (OSError, KeyError, ValueError):
        return None


def tree_batch_roofline(lens, f, contra, pass_s, ms_in, ms_out):
    """The tree-order batch form's line: its HBM utilisation from MEASURED bytes (the slice's PMC record,
    scaled by nothing: bytes and time of the same slice), and the 8d figure of the reference's loops
    as a throughput (never a utilisation: the mode does not move those bytes)."""
    from rna_algos_amd import workloads as W
    b_8d = float(W.algorithmic_bytes(lens, contra, f))
    r = {"kernel": "tree-order batch form: k_tlane_inside / k_tlane_outside (a lane per cell, one diagonal per "
                   "launch), k_tlane_gen (generic 2-loop sums of the listed cells, three diagonals per launch), "
                   "k_tree_mid_mx (mid-field of the cubic products on the matrix cores)",
         "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "ms_inside_per_step": ms_in, "ms_outside_per_step": ms_out,
         "reference_equivalent_throughput": {
             "what": "SURVEY 8d bytes of the REFERENCE's loops for this batch over this pass: the contract's "
                     "figure, not a utilisation (the mode replaces three cubic loops by prefix recurrences "
                     "and matrix-core products)",
             "GB/s": b_8d / pass_s / 1e9, "of_peak": b_8d / pass_s / 1e9 / HBM_PEAK_GBS}}
    t = tree_batch_traffic()
    if t:
        tb, ts, tnt, src = t
        r.update({"achieved": tb / ts / 1e9, "frac": tb / ts / 1e9 / HBM_PEAK_GBS, "traffic": tb,
                  "traffic_source": src,
                  "bytes": f"PMC FETCH_SIZE x 2 + WRITE_SIZE of one pass over a 1 000-sequence slice of this batch "
                           f"({tnt} nt, every 10th sequence) over that slice's pass time: MEASURED bytes, the "
                           f"hardware utilisation of the mode on this workload"})
    else:
        r.update({"achieved": None, "frac": None, "traffic": None,
                  "bytes": "no PMC record of the batch form committed (scripts/prof_traffic_tree_batch.sh)"})
    return r


def tree_leg(ctx, torch, dev, stream, seq, contra, reps, f, ref=None):
    """ms per sequence of ONE sequence in tree-order mode (median of `reps` calls after one
    warm-up call, device-resident), its rooflines, and its deviation from the reference-order
    result `ref` = (d_out, d_logz) of the same sequence."""
    n = len(seq)
    b = torch.from_numpy(np.ascontiguousarray(seq)).to(dev)
    o = torch.empty(n * (n + 1) // 2, dtype=torch.float32, device=dev)
    z = torch.empty(1, dtype=torch.float32, device=dev)
    off = np.array([0, n], dtype=np.uint64)
    oo = np.array([0, n * (n + 1) // 2], dtype=np.uint64)
    ctx.set("summation_mode", 1)
    try:
        ms = []
        for r in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.bpp_batch_device(1, b.data_ptr(), off, contra, False, o.data_ptr(), oo, z.data_ptr(), stream)
            torch.cuda.synchronize()
            if r:
                ms.append((time.perf_counter() - t0) * 1e3)
        st = ctx.stats()
    finally:
        ctx.set("summation_mode", 0)
    med = float(np.median(ms))
    by = tree_bytes(n, f, contra)
    # HBM bytes of the whole sweep from the committed PMC passes (scripts/prof_tree.sh: FETCH_SIZE
    # doubled as the gfx950 guide prescribes, + WRITE_SIZE), n = 4096 Turner only; newest round first
    traffic = traffic_src = None
    if n == 4096 and not contra:
        for name in ("r04_tree_n4096_traffic.json", "r03_tree_n4096_traffic.json"):
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", name)))
                traffic, traffic_src = float(t["total_bytes_x2"]), "profiles/" + name
                break
            except Exception:
                continue
    # The leg's roofline is priced in bytes the mode MOVES: the PMC traffic of the committed pass
    # where there is one, else the bytes its kernels' loads and stores name.  SURVEY 8d's
    # streamed-operand bytes of the REFERENCE's loops (what BASELINE.md section 3 quotes the
    # north_star's ">= 50 % of roofline" in) are not moved by this mode — three of the reference's
    # Theta(n^3) loops are prefix recurrences here and the products are LDS-tiled — so that figure is
    # a throughput EQUIVALENT, reported under its own key and never as a utilisation.
    moved = traffic if traffic is not None else by["inside"] + by["outside"]
    res = {
        "ms_per_seq": med, "ms_all_calls": [round(x, 2) for x in ms],
        "ms_inside": st["ms_inside"], "ms_outside": st["ms_outside"],
        "launches": st["launches_inside"] + st["launches_outside"],
        "summation": "tree order: order-free logsumexp sums, hardware exp2/log2; NOT bit-comparable "
                     "with the reference (include/rnamc.h, rnamc_ctx_set summation_mode)",
        "roofline": {
            "kernel": "tree-order sweep (k_tree_* launches with k_tree_mid / k_tree_ext beside them)",
            "bound": "hbm", "achieved": moved / (med * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": moved / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": traffic_src,
            "bytes": ("PMC FETCH_SIZE x 2 + WRITE_SIZE of the committed pass" if traffic is not None else
                      "bytes the kernels' loads and stores name (no PMC record for this workload)"),
            "note": "hardware utilisation: the sweep is bound by its chain of dependent launches / in-band "
                    "steps, not by bytes (DESIGN.md section 4b)",
        },
        "reference_equivalent_throughput": {
            "what": "SURVEY 8d streamed-operand bytes of the REFERENCE's loops ([16 + 8 + 12 f] T + 12*496 f "
                    "n^2/2 + 36 n^2) divided by this mode's time: the unit BASELINE.md section 3 defines "
                    "and the north_star's '>= 50 % of the HBM roofline at n = 4096' (<= 86 ms) is quoted in. "
                    "NOT a hardware utilisation: this mode does not move those bytes (the figure can exceed 1)",
            "algorithmic_bytes_of_reference_loops": by["b_8d"],
            "GBps_equivalent": by["b_8d"] / (med * 1e-3) / 1e9,
            "ratio_to_8TBps": by["b_8d"] / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
        },
        "roofline_moved": {
            "what": "bytes the tree-order kernels' loads and stores name (the cell-independent "
                    "Theta(n^3) loops of the reference are prefix recurrences here, and two cells "
                    "share one row stream)",
            "bound": "hbm", "bytes": by["inside"] + by["outside"],
            "achieved": (by["inside"] + by["outside"]) / ((st["ms_inside"] + st["ms_outside"]) * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (by["inside"] + by["outside"]) / ((st["ms_inside"] + st["ms_outside"]) * 1e-3) / 1e9 / HBM_PEAK_GBS,
        },
        "roofline_valu": {
            "what": "f32 VALU issue slots counted from the kernels (~10 per product term incl. exp2 at "
                    "quarter rate, ~45 per 2-loop probe) against 256 CUs x 4 SIMD-32 x 2.4 GHz",
            "bound": "valu", "slots": by["valu_slots"], "peak_slots_per_s": VALU_PEAK_LANE_OPS,
            "time_at_peak_ms": by["valu_slots"] / VALU_PEAK_LANE_OPS * 1e3,
            "frac": by["valu_slots"] / VALU_PEAK_LANE_OPS / (med * 1e-3),
        },
    }
    if ref is not None:
        ro, rz = ref
        ka, kb = o >= -0.5, ro >= -0.5
        both = ka & kb
        res["deviation_from_reference_order"] = {
            "key_sets_equal": bool(torch.equal(ka, kb)),
            "max_abs_dp": float((o[both] - ro[both]).abs().max()) if bool(both.any()) else 0.0,
            "d_lnZ": float(z[0] - rz[0]), "lnZ_tree": float(z[0]), "lnZ_reference_order": float(rz[0]),
            "note": "the reference's logsumexp is an approximate, order-dependent fold: this is ITS "
                    "distance from an order-free f32 sum; against the f64 value of the same recurrences "
                    "the tree-order mode is within 1e-4 at n = 1024 (tests/test_gpu_tree.py), the "
                    "reference-order mode 1.2e-2",
        }
    return res


def golden_digest(packed):
    """sha256 of the f32 bits with the libm-exp branch (p >= 0.9999) canonicalised, as
    tests/make_golden.py computes it from the oracle."""
    a = np.array(packed, dtype=np.float32, copy=True)
    a[a >= 0.9999] = 1.0
    return hashlib.sha256(a.tobytes()).hexdigest()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cgroup_cpu_quota():
    """CPUs' worth of time the cgroup grants (cpu.max), or None when unlimited / unreadable"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        return None


def usable_cores():
    """hardware threads this process may run on: its affinity mask, capped by a cgroup CPU quota
    (threads beyond the quota would time the scheduler, not the port)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    q = cgroup_cpu_quota()
    return max(1, min(n, int(q))) if q else n


def cpu_baseline(params, seqs, contra, budget_s):
    """Oracle ("port" of the reference's CPU path) on a bounded sample, one sequence per
    thread on all usable host cores (as the reference's thread pool does).  Every thread gets
    a sequence of (nearly) the SAME cost, so that all cores are busy for the whole wall time:
    the `cores` sequences of the workload whose lengths are closest to the length whose cost
    fits the budget."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from rna_algos_amd import workloads as W
    # all usable host cores, as BASELINE.md section 2 plans and the reference's pool does
    # (num_cpus::get(), src/bin/mccaskill_algo.rs:44-48); RNAMC_CPU_BASELINE_CORES caps it
    cores = usable_cores()
    if os.environ.get("RNAMC_CPU_BASELINE_CORES"):
        cores = max(1, min(cores, int(os.environ["RNAMC_CPU_BASELINE_CORES"])))
    lens = np.array([len(s) for s in seqs])
    # ~60 ns per (cell,k) per core for the dense row-major restatement on one thread per core
    # (EPYC 9575F, round-3 records); with every hardware thread busy a thread gets about half a
    # core and the memory system is shared: budgeted at 150 ns
    n_target = (6.0 * budget_s / 150e-9) ** (1.0 / 3.0)
    if len(seqs) >= cores:
        pick = np.argsort(np.abs(lens - n_target), kind="stable")[:cores]
        sample = [seqs[i] for i in pick]
    else:
        # fewer sequences than cores (a single long sequence): the reference folds one sequence
        # on one thread, so one thread per sequence times a prefix whose cost fits the budget
        m = int(min(n_target, lens.min()))
        sample = [s[:m] for s in seqs]
    t0 = time.time()
    O.bpp_batch(params.ptr, sample, contra, False, n_threads=len(sample), want_bpp=False)
    dt = time.time() - t0
    nt = int(sum(len(s) for s in sample))
    slens = np.array([len(s) for s in sample])
    T = float(W.pair_cost(slens).sum())
    T_all = float(W.pair_cost(lens).sum())
    return {
        # scaled to the metric's unit: the workload's nt/s if the whole of it ran at the
        # sample's rate per (cell,k) iteration (cost model sum n(n^2-1)/6)
        "value": float(lens.sum()) / (dt * T_all / T), "unit": "nt/s",
        "cores": int(len(sample)), "kind": "port",
        "sample": f"{len(sample)} sequences of the workload with lengths {int(slens.min())}.."
                  f"{int(slens.max())} (equal cost per thread), one per thread, {dt:.1f} s wall; "
                  f"extrapolated to the workload by sum n(n^2-1)/6.  The port is faster than the "
                  f"Rust reference would be (dense arrays, no twoloop_scores hash map)",
        "os_cpu_count": os.cpu_count(), "usable_cores": usable_cores(),
        "cgroup_cpu_quota": cgroup_cpu_quota(), "cpu_model": cpu_model(),
        "sample_nt_per_s": nt / dt,
        "ns_per_cell_k_per_core": dt * 1e9 * len(sample) / T,
    }


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def relaunch_under_torchrun(args):
    """--gpus N > 1 without a torchrun environment: start the launcher as a CHILD process
    (nothing in this process has touched the GPU yet) and relay its output."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"bench.py: --gpus {args.gpus} outside torchrun: launching {' '.join(cmd[1:8])} ...",
          file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


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
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="no per-kernel HIP events in the last warm-up pass")
    ap.add_argument("--no-n4096", action="store_true",
                    help="skip the single n=4096 sequence (second half of BASELINE.json's metric)")
    ap.add_argument("--no-tree-batch", action="store_true",
                    help="skip the tree-order pass of the same batch (value_tree, a second figure)")
    ap.add_argument("--no-transfers", action="store_true",
                    help="skip the host-buffer pass (value_with_transfers)")
    ap.add_argument("--time-budget-s", type=float, default=555.0,
                    help="wall-clock budget of the whole run: optional legs (transfers pass, "
                         "n=4096, CPU baseline) are dropped, with a note in the line, when the "
                         "W + K passes would not leave room for them")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--share-devices", action="store_true",
                    help="rehearsal only: rank r uses device r mod device_count, so that N ranks "
                         "can be rehearsed on fewer GPUs (gloo backend; the line says so)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no device work: exercises launcher, sharding, barrier and the "
                         "max-over-ranks reduction (CPU test of the N > 1 path, with gloo)")
    ap.add_argument("--set", action="append", default=[], metavar="KNOB=VALUE",
                    help="rnamc_ctx_set knob (tuning experiments)")
    ap.add_argument("--summation", default="reference", choices=["reference", "tree"],
                    help="tree: the whole workload in the tree-order summation mode — a SECOND figure, never the "
                         "headline (the parity gate is the reference-order mode): the line is labelled, the golden "
                         "members are compared with a reference-order pass of the same run (key sets, max |dp|, "
                         "|d ln Z|) instead of by sha256, the per-kernel rooflines of the reference-order kernels "
                         "and the n = 4096 leg are left out")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.rehearse_shard:
        sys.exit(relaunch_under_torchrun(args))

    import torch
    import torch.distributed as dist
    from rna_algos_amd import workloads as W

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: refusing to report a "
                  f"{world}-rank run as {args.gpus} GPUs", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group("gloo")
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

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world == 1:
            return x
        dev_t = torch.device(f"cuda:{local_rank}") if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([x], dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_ranks(values):
        """every rank's list of floats -> [world][len(values)] on every rank (scalars only: the
        step time, sequence / nucleotide / golden-check counts; never data of the path)"""
        if world == 1:
            return [list(map(float, values))]
        dev_t = torch.device(f"cuda:{local_rank}") if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor(list(map(float, values)), dtype=torch.float64, device=dev_t)
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [[float(v) for v in o.tolist()] for o in out]

    def per_rank_fields(mean_pass_s, checked_here):
        """what a multi-GPU line is judged on: every rank's mean pass time, its share of the batch,
        the cost model's prediction for it and how many golden members it checked"""
        rows = gather_ranks([mean_pass_s, len(my_seqs), float(lens.sum()), float(costs[mine].sum()),
                             checked_here])
        ps = [r[0] for r in rows]
        pc = [r[3] for r in rows]
        return {
            "per_rank_s": [round(x, 6) for x in ps],
            "per_rank_sequences": [int(r[1]) for r in rows],
            "per_rank_nt": [int(r[2]) for r in rows],
            "per_rank_model_cost_s": [round(x, 6) for x in pc],
            "per_rank_golden_checked": [int(r[4]) for r in rows],
            "imbalance": (max(ps) / (sum(ps) / len(ps))) if min(ps) > 0 else None,
            "model_imbalance": (max(pc) / (sum(pc) / len(pc))) if min(pc) > 0 else None,
        }

    total_nt = int(lens_all.sum()) if len(seqs) >= world else int(lens_all.sum()) * world
    base = {
        "metric": "nucleotides/sec (batch)" if args.workload == "batch10k" else "nucleotides/sec",
        "unit": "nt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
    }

    if args.dry_run:
        # the N > 1 plumbing without a device: partition, barrier, max-reduce of a step time
        barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        cnt = len(my_seqs)
        if world > 1:
            c = torch.tensor([cnt, int(lens.sum())], dtype=torch.int64)
            if args.backend == "nccl":
                c = c.to(f"cuda:{local_rank}")
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            cnt, nt_sum = (int(x) for x in c.tolist())
            assert len(seqs) < world or (cnt == len(seqs) and nt_sum == int(lens_all.sum())), \
                "shards do not partition the batch"
        prf = per_rank_fields(0.01 * (rank + 1), 0)
        if rank == 0:
            base.update(prf)
            base.update({"value": 0.0, "ms_per_step": elapsed * 1e3, "dry_run": True,
                         "config": {"workload": label, "sequences_all_ranks": cnt,
                                    "sharding": f"{args.shard}, {world} rank(s), backend "
                                                f"{args.backend}"}})
            print(json.dumps(base), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    from rna_algos_amd.utils import FoldScoreSets
    from rna_algos_amd.mccaskill_algo import Context
    if args.share_devices:
        if args.backend != "gloo":
            print("bench.py: --share-devices needs --backend gloo (RCCL refuses two ranks on one "
                  "device)", file=sys.stderr)
            sys.exit(2)
        local_rank = local_rank % torch.cuda.device_count()
        label += f" [REHEARSAL: {world} ranks share {torch.cuda.device_count()} device(s)]"
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    params = FoldScoreSets.synthetic(args.param_seed)
    ctx = Context(params, device=local_rank)
    ctx.set("profile", 1)
    if args.group_max_seqs:
        ctx.set("group_max_seqs", args.group_max_seqs)
    if args.group_max_nt:
        ctx.set("group_max_nt", args.group_max_nt)
    if args.group_ws_gb:
        ctx.set("group_ws_bytes", args.group_ws_gb << 30)
    for kv in args.set:
        k, v = kv.split("=")
        ctx.set(k, int(v))
    tree = args.summation == "tree"
    if tree:
        ctx.set("summation_mode", 1)
        args.no_kernel_timing = args.no_n4096 = True

    # inputs resident in HBM before the timed region; outputs stay in HBM
    h_bases = np.concatenate(my_seqs)
    d_bases = torch.from_numpy(h_bases).to(dev)
    d_out = torch.empty(int(out_offsets[-1]), dtype=torch.float32, device=dev)
    d_logz = torch.empty(len(my_seqs), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def progress(msg):
        # stderr only (stdout carries the ONE JSON line): a run of many minutes is not silent
        if rank == 0:
            print(f"bench.py [{time.time() - T_START:6.1f} s] {msg}", file=sys.stderr, flush=True)

    def step():
        ctx.bpp_batch_device(len(my_seqs), d_bases.data_ptr(), offsets, contra, False,
                             d_out.data_ptr(), out_offsets, d_logz.data_ptr(), stream)

    notes = []
    ms_main = ms_tail = ms_small = 0.0
    l_main = l_tail = l_small = 0
    with_transfers = None
    pass_s = None  # measured duration of one pass (first warm-up)

    passes_due = args.steps + args.warmup

    def room_for(extra_s):
        """is there room for an optional leg of `extra_s` seconds beside the passes still due?"""
        if pass_s is None:
            return True
        return (time.time() - T_START) + passes_due * pass_s + extra_s <= args.time_budget_s

    # optional legs after the timed steps (n = 4096 once, CPU baseline) and the read-back
    legs_s = 8.0 + (0.0 if args.no_n4096 else 9.0) + (0.0 if args.no_cpu_baseline else args.cpu_budget_s + 4.0)
    warm_done = 0
    w = 0
    tr_wanted = (args.warmup >= 2 and not args.no_transfers and rank == 0 and world == 1
                 and not args.rehearse_shard)
    # the host-entry pass is the SECOND warm-up pass when there are three or more (the first
    # one then has allocated the DP workspace, which is not part of a transfer), else the first
    tr_at = 2 if args.warmup >= 3 else 1
    while w < args.warmup:
        last = w == args.warmup - 1
        if pass_s is not None and not last and warm_done >= (tr_at if tr_wanted else 1):
            # Time budget (a bench killed by the driver's timeout is an unmeasured round): the
            # K timed steps are kept exact as long as possible; warm-up passes beyond the
            # first and the last (which carries the per-kernel timing) are dropped first.
            projected = ((time.time() - T_START) + (args.warmup - w + args.steps) * pass_s + legs_s)
            if max_over_ranks(projected) > args.time_budget_s:
                notes.append(f"time budget {args.time_budget_s:.0f} s ({pass_s:.1f} s per pass): "
                             f"warm-up passes {w + 1}..{args.warmup - 1} of {args.warmup} dropped "
                             f"(kept: the first, the host-entry pass, the per-kernel timing pass)")
                w = args.warmup - 1
                last = True
        passes_due = args.steps + args.warmup - w - 1
        warm_done += 1
        w += 1
        want_tr = tr_wanted and warm_done == tr_at
        if want_tr:
            # first warm-up pass through the host-buffer entry: H2D + kernels + D2H.  Host
            # buffers are allocated and touched before the clock starts (the caller owns them).
            h_out = np.empty(int(out_offsets[-1]), dtype=np.float32)
            h_out[::1024] = 0.0  # touch every page
            h_logz = np.empty(len(my_seqs), dtype=np.float32)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.bpp_batch_into(h_bases, offsets, contra, False, h_out, out_offsets, h_logz)
            dt = time.perf_counter() - t0
            with_transfers = {"value": float(lens.sum()) / dt, "unit": "nt/s", "s": dt,
                              "what": "one pass of rnamc_bpp_batch (pageable host buffers in, "
                                      "host buffers out): H2D + kernels + D2H"}
            pass_s = dt if pass_s is None else min(pass_s, dt)
            del h_out
            progress(f"warm-up pass {w} of {args.warmup} (host-buffer entry): {dt:.2f} s")
            continue
        if last and rank == 0 and not args.no_kernel_timing:
            # per-kernel durations for the roofline: a pair of HIP events around every
            # outside-sweep kernel, on the stream it is launched on (costs ~2 %: warm-up only)
            ctx.set("profile", 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pass_s = dt if pass_s is None else min(pass_s, dt)
        progress(f"warm-up pass {w} of {args.warmup}: {dt:.2f} s")
        if last and rank == 0 and not args.no_kernel_timing:
            st = ctx.stats()
            ctx.set("profile", 1)
            ms_main, ms_tail, ms_small = (st["ms_outside_main"], st["ms_outside_tail"],
                                          st["ms_outside_small"])
            l_main, l_tail, l_small = (st["launches_outside_main"], st["launches_outside_tail"],
                                       st["launches_outside_small"])
    steps = args.steps
    if pass_s is not None:
        # last resort: fewer timed steps than asked for, reported as such
        fit = int((args.time_budget_s - (time.time() - T_START) - 5.0) / pass_s)
        fit = max_over_ranks(float(-fit)) * -1.0  # the smallest fit over ranks
        if fit < steps:
            steps = max(1, int(fit))
            notes.append(f"time budget {args.time_budget_s:.0f} s: timed {steps} of the "
                         f"{args.steps} steps asked for ({pass_s:.1f} s per pass)")
    torch.cuda.synchronize()
    barrier()
    ms_in = ms_out = 0.0
    l_in = l_out = 0
    t0 = time.perf_counter()
    step_s = []
    for _ in range(steps):
        ts = time.perf_counter()
        step()
        st = ctx.stats()  # event-timed sweeps of this step (the call synchronised its stream)
        step_s.append(time.perf_counter() - ts)
        progress(f"timed step {len(step_s)} of {steps}: {step_s[-1]:.2f} s")
        ms_in += st["ms_inside"]
        ms_out += st["ms_outside"]
        l_in += st["launches_inside"]
        l_out += st["launches_outside"]
    torch.cuda.synchronize()
    barrier()
    local_elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(local_elapsed)
    pass_s = elapsed / max(steps, 1)
    passes_due = 0

    # parity, outside the timed region: batch members with committed oracle checksums
    checked = failed = 0
    tree_dev = None
    def tree_deviation():
        """d_out / d_logz hold a TREE-order pass: the golden members once more in REFERENCE order (checked
        against the oracle's sha256) and the tree-order result's distance from that
        -> (checked, failed, record)"""
        n_checked = n_failed = 0
        gold = {}
        for name in ("checksums_batch.json", "checksums_batch2k.json"):
            try:
                gold.update(json.load(open(os.path.join(ROOT, "tests", "golden", name)))["cases"])
            except OSError:
                pass
        where = {int(g): x for x, g in enumerate(mine)}
        picks = []
        for key, info in gold.items():
            idx = int(key.split("_")[0][len("batch"):])
            if key.endswith("contra") == contra and idx in where and idx < args.batch_count:
                picks.append((where[idx], info))
        if not picks:
            return 0, 0, None
        ctx.set("summation_mode", 0)
        try:
            rmats, rz = ctx.bpp_batch([my_seqs[x] for x, _ in picks], contra, False)
        finally:
            ctx.set("summation_mode", 1)
        worst_p = worst_z = 0.0
        keys_equal = True
        for (x, info), rm, z0 in zip(picks, rmats, rz):
            n_checked += 1
            ref = np.asarray(rm.packed)
            if golden_digest(ref) != info["sha256"]:
                n_failed += 1
            got = d_out[int(out_offsets[x]):int(out_offsets[x + 1])].cpu().numpy()
            ka, kb = got >= -0.5, ref >= -0.5
            keys_equal = keys_equal and bool(np.array_equal(ka, kb))
            both = ka & kb
            worst_p = max(worst_p, float(np.abs(got[both].astype(np.float64) - ref[both]).max()))
            worst_z = max(worst_z, abs(float(d_logz[x]) - float(z0)))
        rec = {"members": len(picks), "key_sets_equal": keys_equal, "max_abs_dp": worst_p,
               "max_abs_d_lnZ": worst_z,
               "note": "against the reference-order result of the same sequences in the same run (itself "
                       "sha256-identical to the oracle); the reference's fold is approximate: this is ITS "
                       "distance from an order-free f32 sum (f64 fixtures: tests/test_gpu_tree.py)"}
        if not keys_equal:
            n_failed += 1
        return n_checked, n_failed, rec

    if tree and args.workload == "batch10k" and args.param_seed == 1:
        checked, failed, tree_dev = tree_deviation()
    if not tree and args.workload == "batch10k" and args.param_seed == 1:
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "checksums_batch.json")))["cases"]
        except OSError:
            gold = {}
        where = {int(g): x for x, g in enumerate(mine)}
        for key, info in gold.items():
            idx = int(key.split("_")[0][len("batch"):])
            if key.endswith("contra") != contra or idx not in where or idx >= args.batch_count:
                continue
            x = where[idx]
            got = d_out[int(out_offsets[x]):int(out_offsets[x + 1])].cpu().numpy()
            checked += 1
            if golden_digest(got) != info["sha256"] or len(my_seqs[x]) != info["n"]:
                failed += 1
    probe = d_out[: min(d_out.numel(), 1 << 22)]
    pres = probe[probe >= -0.5]
    assert pres.numel() > 0 and float(pres.min()) >= -0.001 and float(pres.max()) < 1.05
    assert bool(torch.isfinite(d_logz).all())
    # also the golden member of the 2048-nt class (tests/golden/checksums_batch2k.json)
    if not tree and args.workload == "batch10k" and args.param_seed == 1 and args.batch_count == 10000:
        try:
            gold2 = json.load(open(os.path.join(ROOT, "tests", "golden", "checksums_batch2k.json")))["cases"]
        except OSError:
            gold2 = {}
        where = {int(g): x for x, g in enumerate(mine)}
        for key, info in gold2.items():
            idx = int(key.split("_")[0][len("batch"):])
            if key.endswith("contra") != contra or idx not in where:
                continue
            x = where[idx]
            got = d_out[int(out_offsets[x]):int(out_offsets[x + 1])].cpu().numpy()
            checked += 1
            if golden_digest(got) != info["sha256"] or len(my_seqs[x]) != info["n"]:
                failed += 1
    # (a rank's OWN pass time: its timed steps, each synchronised; `local_elapsed` ends behind the barrier
    # and so is the slowest rank's time on every rank)
    prf = per_rank_fields(float(np.mean(step_s)) if step_s else local_elapsed / max(steps, 1), checked)
    if world > 1:
        # every rank checks the golden members of ITS shard; rank 0 reports the sum
        cf = torch.tensor([checked, failed], dtype=torch.int64,
                          device=dev if args.backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(cf, op=dist.ReduceOp.SUM)
        checked, failed = (int(x) for x in cf.tolist())
    if failed:
        if rank == 0:
            print(f"bench.py: {failed} of {checked} golden batch members differ from the oracle's "
                  f"checksum", file=sys.stderr)
        sys.exit(3)

    if rank == 0:
        f = float(np.mean([W.paired_fraction(s) for s in my_seqs[:: max(1, len(my_seqs) // 8)]]))
        lf = lens.astype(np.float64)
        b_out = outside_bytes(lf, f)
        total_T = float(W.pair_cost(lf).sum())
        b_main = outside_bytes(lf, f, "main")
        b_tail = outside_bytes(lf, f, "tail")
        if l_main == 0:
            # nothing was large enough to split (single sequences): all roles ran in
            # k_outside<.,7>, which then is the dominant kernel
            b_main, ms_main, l_main = b_out, ms_small, l_small
        b_in = inside_bytes(lf, f, contra, "two_diagonal")
        b_in_model = inside_bytes(lf, f, contra, "model")
        ach_out = b_out * steps / (ms_out * 1e-3) / 1e9 if ms_out > 0 else 0.0
        ach_in = b_in * steps / (ms_in * 1e-3) / 1e9 if ms_in > 0 else 0.0

        def roof(kernel, b, ms, launches, pmc_key, **extra):
            ach = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            traffic, src = (pmc_traffic_per_launch(pmc_key, total_T, launches, contra)
                            if args.workload == "batch10k" and pmc_key else (None, None))
            r = {"kernel": kernel, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                 "traffic_source": src, "algorithmic_bytes_per_launch": b / max(launches, 1),
                 "avg_launch_ms": ms / max(launches, 1), "launches_per_step": launches}
            r.update(extra)
            return r

        res = dict(base)
        res.update(prf)
        res.update({
            "value": total_nt * steps / elapsed,
            "steps": steps,
            "ms_per_step": elapsed * 1e3 / steps,
            "config": {
                "workload": label,
                "model": "Turner-2004-shaped synthetic tables" if not contra
                         else "CONTRAfold-shaped synthetic tables",
                "tables": f"synthetic seed {args.param_seed} (real tables live in the absent "
                          f"rna-ss-params crate)",
                "summation": ("reference-order (bit-faithful to the CPU path)" if not tree else
                              "TREE ORDER (rnamc_ctx_set summation_mode 1): order-free logsumexp sums, hardware "
                              "exp2 / log2 — NOT bit-comparable with the reference, NOT the parity gate; this line is "
                              "a second figure, the headline is the reference-order run of the same command"),
                "allows_short_hairpins": False,
                "sharding": f"{args.shard} over the measured cost model a*n(n^2-1)/6 + b*n^2, "
                            f"{world} rank(s), no data-path collective",
                "sequences_rank0": len(my_seqs),
                "paired_fraction_f": f,
            },
            "parity_check": ((f"{checked}/{checked} golden members (sha256 of the whole matrix "
                              f"against the oracle's, tests/golden/checksums_batch*.json)") if not tree else
                             (f"tree order: {checked} golden members re-run in reference order (sha256-identical to "
                              f"the oracle) and compared: see deviation_from_reference_order"))
            if checked else "none of the golden members is in this run",
            "value_with_transfers": with_transfers["value"] if with_transfers else None,
            "with_transfers": with_transfers,
            # the dominant kernel by GPU time (rocprofv3 --stats): per-kernel accounting, HIP
            # events around each of its launches on its own stream, taken in the last warm-up
            # pass.  Its few small-launch siblings (k_outside<.,7>, < 1 % of the time) do the
            # same roles on the first diagonals; their bytes are left in
            "roofline": roof(
                ("k_outside<.,5> (probs_multibranch + 2-loop half of the pair probabilities, one "
                 "launch per anti-diagonal)") if l_tail else
                ("k_outside<.,7> (outside sweep, all roles, one launch per anti-diagonal)"
                 if args.workload == "batch10k" else
                 "k_outside_lat (latency form of a lone sequence: pair-probability chains, "
                 "probs_multibranch and the 2-loop halves of one anti-diagonal in one launch)"),
                b_main, ms_main, l_main, "k_outside_main",
                note=("runs beside the pair-probability kernels (other streams): they share the chip"
                      if args.workload == "batch10k" else
                      "bound by the length of the dependent fold chains (3(n-d) steps per "
                      "anti-diagonal), not by bytes: DESIGN.md section 4, latency forms")),
            "roofline_tail": roof("k_outside<.,2> (multibranch half of the pair probabilities)",
                                  b_tail, ms_tail, l_tail, "k_outside_tail"),
            "roofline_outside_sweep": {
                "what": "all outside kernels together: algorithmic bytes of the whole outside "
                        "sweep over its event-timed duration",
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
                "traffic": pmc_traffic_per_launch("k_inside", total_T, l_in / steps, contra)[0]
                if args.workload == "batch10k" else None,
                "avg_launch_ms": ms_in / max(l_in, 1), "launches_per_step": l_in // steps,
                "ms_per_step": ms_in / steps,
                "note": "bytes of the two-diagonal schedule (one lane folds two cells off one stream "
                        "of the row operands: 6 B per (cell,k)); by SURVEY 8d's streamed-operand "
                        "model (16 B per (cell,k)) the same sweep scores model_achieved",
                "model_achieved": b_in_model * steps / (ms_in * 1e-3) / 1e9 if ms_in > 0 else 0.0,
            },
        })
        if tree:
            res["summation_mode"] = 1
            res["deviation_from_reference_order"] = tree_dev
            for k in ("roofline_tail", "roofline_outside_sweep", "roofline_inside"):
                res.pop(k, None)
            res["roofline"] = tree_batch_roofline(lens.astype(np.float64), f, contra, elapsed / max(steps, 1),
                                                  ms_in / max(steps, 1), ms_out / max(steps, 1))
        if args.workload != "batch10k":
            # ms per sequence: median over the timed steps (>= 5 after a warm-up)
            res["ms_per_seq"] = float(np.median(step_s)) * 1e3
            res["ms_per_seq_all_steps"] = [round(x * 1e3, 2) for x in step_s]
            res["ms_inside_per_step"] = ms_in / steps
            res["ms_outside_per_step"] = ms_out / steps
            if len(my_seqs) == 1:
                # second summation mode on the same sequence, both reported
                tl = tree_leg(ctx, torch, dev, stream, my_seqs[0], contra, max(steps, 3),
                              W.paired_fraction(my_seqs[0]), (d_out, d_logz))
                res["ms_per_seq_tree"] = tl["ms_per_seq"]
                res["tree"] = tl
        elif world == 1 and not args.no_n4096 and not args.rehearse_shard:
            # the other half of the metric: ms per sequence at n = 4096 (BASELINE.json
            # configs[2]: Turner), ONE timed call, device-resident (the median of >= 5 after a
            # warm-up is `--workload n4096 --steps 5 --warmup 1`, profiles/)
            if room_for(6.0 + (0 if args.no_cpu_baseline else args.cpu_budget_s + 3)):
                s4 = W.synthetic_seq(4096, 4096)
                b4 = torch.from_numpy(s4).to(dev)
                o4 = torch.empty(4096 * 4097 // 2, dtype=torch.float32, device=dev)
                z4 = torch.empty(1, dtype=torch.float32, device=dev)
                off4 = np.array([0, 4096], dtype=np.uint64)
                oo4 = np.array([0, 4096 * 4097 // 2], dtype=np.uint64)
                # median of 3 calls when the budget has room for them, else one call (said so)
                calls4 = 3 if room_for(12.0 + (0 if args.no_cpu_baseline else args.cpu_budget_s + 3)) else 1
                ms4 = []
                for _ in range(calls4):
                    torch.cuda.synchronize()
                    t4 = time.perf_counter()
                    ctx.bpp_batch_device(1, b4.data_ptr(), off4, contra, False, o4.data_ptr(), oo4,
                                         z4.data_ptr(), stream)
                    torch.cuda.synchronize()
                    ms4.append((time.perf_counter() - t4) * 1e3)
                res["ms_per_seq_n4096"] = float(np.median(ms4))
                res["ms_per_seq_n4096_calls"] = [round(x, 1) for x in ms4]
                st4 = ctx.stats()
                # the same sequence in tree-order mode: the north_star's ">= 50 % of the HBM
                # roofline at n = 4096" is only reachable there
                tl = tree_leg(ctx, torch, dev, stream, s4, contra, 3, W.paired_fraction(s4), (o4, z4))
                res["ms_per_seq_n4096_tree"] = tl["ms_per_seq"]
                res["n4096_tree"] = tl
                res["n4096_note"] = (
                    f"single n=4096 sequence, same tables, reference-order mode, median of "
                    f"{calls4} call(s) (inside {st4['ms_inside']:.0f} ms"
                    f", outside {st4['ms_outside']:.0f} ms): a lock-step group of one is bound by "
                    f"the sequential fold chains the reference's summation order dictates "
                    f"(n^2/2 = 8.4 M dependent steps inside, 3 n^2/2 = 25 M outside), not by HBM")
            else:
                notes.append("n=4096 leg skipped: time budget")
        if not args.no_cpu_baseline and world == 1:
            if room_for(args.cpu_budget_s + 3):
                res["cpu_baseline"] = cpu_baseline(params, my_seqs, contra, args.cpu_budget_s)
            else:
                notes.append("cpu_baseline leg skipped: time budget")
                res["cpu_baseline"] = None
        if (not tree and args.workload == "batch10k" and world == 1 and not args.rehearse_shard
                and not args.no_tree_batch and args.param_seed == 1):
            # the SAME batch in the tree-order summation mode, a second figure (never `value`): one warm-up
            # pass, one timed pass, the golden members compared with their reference-order results.  Last
            # leg, and only when the time budget has room for it.
            est = pass_s / 2.5
            # (a dozen seconds of margin on top: the leg must not be what pushes a run past the driver's limit)
            if room_for(2.0 * est + 8.0 + 12.0):
                ctx.set("summation_mode", 1)
                try:
                    tt = []
                    for _ in range(2):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        step()
                        torch.cuda.synchronize()
                        tt.append(time.perf_counter() - t0)
                    stt = ctx.stats()
                    progress(f"tree-order leg: {tt[0]:.2f} s (warm-up), {tt[1]:.2f} s")
                    n_c, n_f, dev_rec = tree_deviation()
                finally:
                    ctx.set("summation_mode", 0)
                res["value_tree"] = float(lens.sum()) / tt[1]
                res["tree_batch"] = {
                    "what": "the same batch, device-resident, in the tree-order summation mode (rnamc_ctx_set "
                            "summation_mode 1): order-free logsumexp sums, hardware exp2 / log2 — NOT bit-comparable "
                            "with the reference and NOT the parity gate; `value` above is the reference-order run",
                    "value": float(lens.sum()) / tt[1], "unit": "nt/s", "s_per_pass": tt[1], "s_warmup_pass": tt[0],
                    "passes": "one warm-up, one timed",
                    "deviation_from_reference_order": dev_rec,
                    "reference_order_members_sha256_ok": n_c - n_f if dev_rec else None,
                    "roofline": tree_batch_roofline(lf, f, contra, tt[1], stt["ms_inside"], stt["ms_outside"]),
                }
                if n_f:
                    notes.append(f"tree-order leg: {n_f} of {n_c} golden members failed their check")
            else:
                notes.append("tree-order batch leg skipped: time budget")
        if steps != args.steps:
            res["steps_requested"] = args.steps
        if warm_done != args.warmup:
            res["warmup"] = warm_done
            res["warmup_requested"] = args.warmup
        if notes:
            res["notes"] = notes
        res["wall_s"] = time.time() - T_START
        print(json.dumps(res), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
