"""CPU suite, part 2: the C-ABI library loads and exports every symbol include/rnamc.h
declares, host-side logic (encoding, parameter plumbing, centroid fold, sharding) works,
and the product refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    from rna_algos_amd import _lib
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "rnamc.h")).read()
    # prototypes start a line with a return type: "int rnamc_x(", "const char* rnamc_y(" ...
    declared = set(re.findall(r"^(?:const )?[a-z_0-9]+\*? (rnamc_[a-z0-9_]+)\(", hdr, re.M))
    assert declared, "no prototypes found"
    for sym in sorted(declared):
        assert hasattr(L, sym), f"librnamc.so lacks {sym}"
    assert set(_lib.SYMBOLS) == declared
    assert L.rnamc_abi_version() == 3
    assert L.rnamc_params_sizeof() > 300000


def test_bytes2seq(built):
    from rna_algos_amd.utils import bytes2seq
    from rna_algos_amd import _lib
    assert list(bytes2seq(b"ACGUacgu")) == [0, 1, 2, 3, 0, 1, 2, 3]
    with pytest.raises(_lib.RnamcError):  # the reference panics (src/utils.rs:570-572)
        bytes2seq(b"ACGT")
    assert bytes2seq(b"").shape == (0,)


def test_packed_index(built):
    from rna_algos_amd import _lib
    from rna_algos_amd.mccaskill_algo import bpp_index, bpp_len
    L = _lib.lib()
    n = 37
    seen = set()
    for i in range(n):
        for j in range(i, n):
            k = L.rnamc_bpp_index(n, i, j)
            assert k == bpp_index(n, i, j)
            seen.add(k)
    assert seen == set(range(bpp_len(n))) and L.rnamc_bpp_len(n) == bpp_len(n)


def test_fold_score_sets_plumbing(built, tmp_path):
    from rna_algos_amd.utils import FoldScoreSets
    src = FoldScoreSets.synthetic(7)
    fss = FoldScoreSets.new(0.0)
    assert float(np.abs(fss.stack_scores).max()) == 0.0
    fss.transfer(src)
    canon = lambda a, b: (a + b == 3) or (a + b == 5)
    for a in range(4):
        for b in range(4):
            for c in range(4):
                for d in range(4):
                    v = fss.stack_scores[a, b, c, d]
                    if canon(a, b) and canon(c, d):
                        assert v != 0.0
                    else:  # transfer() leaves non-canonical entries at init_val
                        assert v == 0.0
                    if not canon(a, b):
                        assert fss.terminal_mismatch_scores[a, b, c, d] == 0.0
    # accumulate(): running f32 sums
    acc = np.float32(0)
    for k in range(31):
        acc = np.float32(acc + fss.hairpin_scores_len[k])
        assert fss.hairpin_scores_len_cumulative[k] == acc
    # synthetic() is deterministic and seed-sensitive
    assert np.array_equal(FoldScoreSets.synthetic(7)._buf, src._buf)
    assert not np.array_equal(FoldScoreSets.synthetic(8)._buf, src._buf)
    # table file round trip
    path = os.path.join(tmp_path, "t.bin")
    src.save(path)
    assert np.array_equal(FoldScoreSets.load(path)._buf, src._buf)
    with open(path, "r+b") as fh:
        fh.write(b"XXXX")
    from rna_algos_amd import _lib
    with pytest.raises(_lib.RnamcError):
        FoldScoreSets.load(path)


def test_no_gpu_means_no_result(built, params):
    """The product path must fail loudly without a HIP device (this container has none)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rna_algos_amd import _lib
    from rna_algos_amd.mccaskill_algo import Context
    with pytest.raises(_lib.RnamcError) as e:
        Context(params)
    assert e.value.status == _lib.ERR_NO_DEVICE


def test_centroid_fold_matches_oracle(built, params, trnas):
    """src/centroid_fold.rs:25-105 driven off bpp matrices, gamma = 2^-7 .. 2^10
    (src/bin/centroid_fold.rs:9-10): product (C++) vs oracle (C), identical structures."""
    from rna_algos_amd.centroid_fold import centroid_fold, get_fold_str, MIN_POW_2, MAX_POW_2
    from rna_algos_amd.mccaskill_algo import BppMatrix
    g = np.load(os.path.join(ROOT, "tests", "golden", "trna_bpp_synthetic_seed1.npz"))
    nonempty = 0
    for idx, (_, s) in enumerate(trnas):
        n = len(s)
        bpp = BppMatrix(n, g[f"trna{idx}_contra"])
        for k in range(MIN_POW_2, MAX_POW_2 + 1):
            gamma = float(2.0 ** k)
            fold = centroid_fold(bpp, n, gamma)
            ref_pairs, ref_acc = O.centroid_fold(bpp.packed, n, gamma)
            assert fold.basepair_pos_pairs == ref_pairs
            assert np.float32(fold.expect_accuracy) == np.float32(ref_acc)
            st = get_fold_str(fold, n)
            assert len(st) == n and st.count("(") == st.count(")") == len(ref_pairs)
            nonempty += bool(ref_pairs)
            # pairs are nested (no crossing) and each has a probability
            for (i, j) in ref_pairs:
                assert bpp[(i, j)] >= -0.5
    assert nonempty > 20


def test_lpt_sharding_properties():
    import bench
    rng = np.random.default_rng(1)
    costs = rng.integers(256, 2049, 1000).astype(np.float64) ** 3
    for world in (1, 2, 4, 8):
        shards = bench.shard_lpt(costs, world)
        allidx = np.concatenate(shards)
        assert sorted(allidx.tolist()) == list(range(1000))
        loads = np.array([costs[s].sum() for s in shards])
        assert loads.max() / loads.mean() < 1.01  # LPT balance on this distribution
        bands = bench.shard_banded(costs, world)
        assert sorted(np.concatenate(bands).tolist()) == list(range(1000))
        loads = np.array([costs[s].sum() for s in bands])
        assert loads.max() / loads.mean() < 1.03
        # bands are contiguous in cost: rank r's cheapest unit is no cheaper than rank r+1's dearest
        for a, b in zip(bands, bands[1:]):
            assert costs[a].min() >= costs[b].max()


def test_workload_generators(built):
    from rna_algos_amd import workloads as W
    assert np.array_equal(W.synthetic_seq(300, 4096), O.splitmix_seq(300, 4096))
    lens = W.batch_lengths()
    assert lens.min() >= 256 and lens.max() <= 2048 and lens.shape == (10000,)
    assert abs(lens.sum() - 1.1444e7) < 1e4  # ~1.15e7 nt
    assert abs(W.pair_cost(lens).sum() / 4.04e12 - 1) < 0.01
    f = W.paired_fraction(W.synthetic_seq(1024, 1024))
    assert 0.35 < f < 0.40


def build_cpp_mirror(tmp_path):
    import subprocess
    exe = os.path.join(tmp_path, "cpp_host_mirror")
    libdir = os.path.join(ROOT, "rna_algos_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp_host_mirror.cpp"), "-L" + libdir, "-lrnamc",
                           "-Wl,-rpath," + libdir])
    return exe


def test_cpp_host_mirror_compiles_and_refuses_without_gpu(built, tmp_path):
    """include/rna_algos/mccaskill_algo.hpp (the C++ mirror of the crate's API) builds with
    plain g++ against the C ABI; without a GPU the program reports 'no device' (exit 77)."""
    import subprocess
    import torch
    exe = build_cpp_mirror(str(tmp_path))
    rc = subprocess.call([exe, os.path.join(ROOT, "tests", "golden", "sampled_trnas.fa")])
    assert rc == (0 if torch.cuda.is_available() else 77)


def test_cli_text_formatting(built):
    """Rust `{}` formatting of f32 (shortest round-trip, positional) used by the binaries'
    text formats (src/bin/mccaskill_algo.rs:104-113, src/bin/centroid_fold.rs:138,180-191)."""
    from rna_algos_amd.bin.mccaskill_algo import fmt_f32, probs2str, HEADER
    from rna_algos_amd.mccaskill_algo import BppMatrix, bpp_index, bpp_len
    assert fmt_f32(1.0) == "1" and fmt_f32(0.5) == "0.5" and fmt_f32(2.0 ** -7) == "0.0078125"
    assert fmt_f32(1024.0) == "1024" and fmt_f32(np.float32(0.1)) == "0.1"
    assert fmt_f32(np.float32(1e-7)) == "0.0000001" and fmt_f32(np.float32(1) / 3) == "0.33333334"
    n = 7
    packed = np.full(bpp_len(n), -1.0, dtype=np.float32)
    packed[bpp_index(n, 1, 6)] = 0.25
    packed[bpp_index(n, 0, 5)] = np.float32(0.1)
    assert probs2str(BppMatrix(n, packed)) == "0,5,0.1 1,6,0.25 "
    assert HEADER.startswith("# Format = >{RNA sequence id}")


def test_params_setters_for_foreign_hosts(built):
    """rnamc_params_set_special_hairpins / _set_hairpin_limits: what a host language that
    does not mirror the struct (the Rust shim) uses for the non-float members."""
    import ctypes as C
    from rna_algos_amd import _lib
    from rna_algos_amd.utils import FoldScoreSets
    L = _lib.lib()
    P = FoldScoreSets.synthetic(3)
    seqs = np.zeros((2, 16), dtype=np.uint8)
    seqs[0, :6] = [1, 0, 0, 0, 0, 2]
    seqs[1, :5] = [2, 3, 3, 3, 1]
    lens = np.array([6, 5], dtype=np.uint8)
    scores = np.array([1.5, -0.25], dtype=np.float32)
    assert L.rnamc_params_set_special_hairpins(P.ptr, 2, seqs.ctypes.data, lens.ctypes.data,
                                               scores.ctypes.data) == 0
    assert L.rnamc_params_set_hairpin_limits(P.ptr, 3, 9, 10) == 0
    # the oracle reads the same block: the planted hairpin's score is picked up
    seq = np.array([1, 0, 0, 0, 0, 2], dtype=np.uint8)
    hp, _, _, _ = O.fold_scores(P.ptr, seq, 0, 0)
    assert hp[len(seq) * (len(seq) + 1) // 2 - 1] == np.float32(1.5)
    # rejected inputs
    bad = lens.copy()
    bad[0] = 17
    assert L.rnamc_params_set_special_hairpins(P.ptr, 2, seqs.ctypes.data, bad.ctypes.data,
                                               scores.ctypes.data) != 0
    assert L.rnamc_params_set_special_hairpins(P.ptr, 65, seqs.ctypes.data, lens.ctypes.data,
                                               scores.ctypes.data) != 0
    assert L.rnamc_params_set_hairpin_limits(P.ptr, 3, 31, 10) != 0


def test_shard_plan_partitions_and_balances(built):
    """rnamc_shard_plan (the partition behind rnamc_bpp_batch_multi): every sequence lands in
    exactly one shard, shards are bands of the length-sorted batch, loads under the cost model
    are level, and the result equals bench.py's shard_banded on the 10k-batch lengths."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from rna_algos_amd import workloads as W
    from rna_algos_amd.mccaskill_algo import shard_plan
    lens = W.batch_lengths(10000)
    costs = W.sweep_cost(lens)
    for world in (1, 2, 3, 4, 8):
        plan = shard_plan(lens, world)
        assert plan.shape == (10000,) and plan.min() == 0 and plan.max() == world - 1
        want = np.empty(10000, dtype=np.uint32)
        for r, idx in enumerate(bench.shard_banded(costs, world)):
            want[idx] = r
        assert np.array_equal(plan, want)
        loads = np.array([costs[plan == r].sum() for r in range(world)])
        assert loads.max() / loads.min() < 1.01
        # bands: every member of shard r is at least as long as every member of shard r+1
        for r in range(world - 1):
            assert lens[plan == r].min() >= lens[plan == r + 1].max()
    # a few sequences of very different cost: one per shard, none left empty
    assert shard_plan([100, 300, 200], 3).tolist() == [2, 0, 1]
    assert sorted(shard_plan([2000, 300, 250, 200, 100], 4).tolist()) == [0, 1, 2, 3, 3]
    for world in (2, 3, 5):
        small = [900, 50, 40, 30, 20, 10, 5]
        want = np.empty(len(small), dtype=np.uint32)
        for r, idx in enumerate(bench.shard_banded(W.sweep_cost(np.array(small)), world)):
            want[idx] = r
        assert np.array_equal(shard_plan(small, world), want)
    assert shard_plan([77], 4).tolist() == [0]
    assert shard_plan([], 4).shape == (0,)


def test_cost_model_has_one_source(built):
    """rnamc_shard_plan, bench.py's shards and workloads.sweep_cost evaluate ONE cost model
    (rnamc_sweep_cost, constants RNAMC_COST_* of include/rnamc.h): equal to the bit with the
    header's formula on the 10k lengths, and the library's plan equals bench.py's banded shards."""
    import re
    import bench
    from rna_algos_amd import _lib, workloads as W
    hdr = open(os.path.join(ROOT, "include", "rnamc.h")).read()
    a = float(re.search(r"#define RNAMC_COST_S_PER_CELL_K\s+(\S+)", hdr).group(1))
    b = float(re.search(r"#define RNAMC_COST_S_PER_N2\s+(\S+)", hdr).group(1))
    lens = W.batch_lengths(10000)
    x = lens.astype(np.float64)
    assert np.array_equal(W.sweep_cost(lens), a * (x * (x * x - 1.0) / 6.0) + b * x * x)
    assert W.sweep_cost(1024) == a * (1024.0 * (1024.0 ** 2 - 1.0) / 6.0) + b * 1024.0 ** 2
    offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    for world in (2, 8):
        shard_of = np.zeros(len(lens), dtype=np.uint32)
        _lib.check(_lib.lib().rnamc_shard_plan(len(lens), offsets.ctypes.data, world, shard_of.ctypes.data))
        for r, idx in enumerate(bench.shard_banded(W.sweep_cost(lens), world)):
            assert np.all(shard_of[idx] == r)
