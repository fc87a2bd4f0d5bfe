"""SURVEY.md 8f-4 on the GPU: rnamc_durbin_batch (anti-diagonal forward / backward kernels)
against the CPU oracle's restatement of src/durbin_algo.rs — bit-identical f32 for every
probability the reference evaluates with its cubic expf (<= 1 ulp where it calls libm exp)."""
import itertools
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def same_bits(got, want, what):
    assert got.shape == want.shape, what
    libm = want >= 0.9999
    assert np.array_equal(got[~libm].view(np.uint32), want[~libm].view(np.uint32)), \
        f"{what}: {int(np.sum(got[~libm] != want[~libm]))} entries differ"
    if libm.any():
        ulp = np.abs(got[libm].view(np.int32).astype(np.int64) - want[libm].view(np.int32))
        assert ulp.max() <= 1, what


def test_all_pairs_of_the_fixture_bit_exact(trnas):
    """the workload of tests/tests.rs:45-80 and benches/benches.rs: all 15 pairs of the 6 tRNAs"""
    from rna_algos_amd.durbin_algo import AlignScores, durbin_algo, durbin_algo_batch, with_pseudo_bases
    s = AlignScores.new(0.0)
    s.transfer()
    seqs = [with_pseudo_bases(x) for _, x in trnas]
    pairs = list(itertools.combinations(range(len(seqs)), 2))
    mats = durbin_algo_batch(seqs, pairs, s)
    assert len(mats) == 15
    for (a, b), m in zip(pairs, mats):
        want = O.durbin(s.ptr, seqs[a], seqs[b])
        same_bits(m, want, f"pair {a},{b}")
        assert m.min() >= -0.001 and m.max() < 1.001  # the reference's own assertion
    one = durbin_algo((seqs[4], seqs[1]), s)
    same_bits(one, O.durbin(s.ptr, seqs[4], seqs[1]), "single pair entry")


def test_ragged_pairs_and_other_scores_bit_exact():
    """lengths from the empty sequence (two pseudo bases) to beyond one workgroup's 1024
    lanes per diagonal, random and zero score sets, a pair of a sequence with itself"""
    from rna_algos_amd.durbin_algo import AlignScores, durbin_algo_batch, with_pseudo_bases
    rng = np.random.default_rng(11)
    lens = [0, 1, 2, 3, 17, 64, 65, 200, 1100, 1300]
    seqs = [with_pseudo_bases(rng.integers(0, 4, n)) for n in lens]
    pairs = [(0, 0), (0, 1), (1, 0), (1, 1), (2, 3), (3, 7), (4, 4), (5, 6), (6, 5), (7, 4),
             (8, 9), (9, 8), (8, 2), (0, 9)]
    sets = []
    s = AlignScores.new(0.0)
    s.transfer()
    sets.append(s)
    sets.append(AlignScores.new(0.0))
    r = AlignScores.new(0.0)
    for name in ("match2match_score", "match2insert_score", "insert_extend_score",
                 "init_match_score", "init_insert_score"):
        r.set(name, rng.uniform(-2, 2))
    r.set("insert_scores", rng.uniform(-1, 1, 4).astype(np.float32))
    r.set("match_scores", rng.uniform(-1, 1, (4, 4)).astype(np.float32))
    sets.append(r)
    for k, sc in enumerate(sets):
        mats = durbin_algo_batch(seqs, pairs, sc)
        for (a, b), m in zip(pairs, mats):
            same_bits(m, O.durbin(sc.ptr, seqs[a], seqs[b]), f"scores {k} pair {a},{b} "
                      f"({len(seqs[a])} x {len(seqs[b])})")


def test_durbin_cli_format(trnas, tmp_path):
    """src/bin/durbin_algo.rs:76-91: header, `>{id1},{id2}`, `i,j,p ` triples of p > 0"""
    from rna_algos_amd.bin import durbin_algo as cli
    from rna_algos_amd.durbin_algo import AlignScores, with_pseudo_bases
    from rna_algos_amd.utils import EXAMPLE_FASTA_FILE_PATH
    out = os.path.join(tmp_path, "match.dat")
    assert cli.main(["-i", EXAMPLE_FASTA_FILE_PATH, "-o", out]) == 0
    text = open(out).read()
    assert text.startswith(cli.HEADER + "\n\n>0,1\n")
    blocks = text[len(cli.HEADER):].split("\n\n>")[1:]
    assert len(blocks) == 15
    s = AlignScores.new(0.0)
    s.transfer()
    seqs = [with_pseudo_bases(x) for _, x in trnas]
    head, body = blocks[7].split("\n", 1)
    a, b = (int(x) for x in head.split(","))
    want = O.durbin(s.ptr, seqs[a], seqs[b])
    got = {}
    for tok in body.split():
        i, j, p = tok.split(",")
        got[(int(i) + 1, int(j) + 1)] = np.float32(p)
    ii, jj = np.nonzero(want > 0)
    assert set(got) == set(zip(ii.tolist(), jj.tolist()))
    assert all(got[(i, j)] == want[i, j] for i, j in got)


def test_durbin_more_pairs_than_one_launch_holds():
    """all n(n-1)/2 pairs of a FASTA of many short records (what src/bin/durbin_algo.rs:55-75
    submits): more than 65535 pairs, the grid-y limit of the match-probability kernel — the
    batch entry must cut its chunks there as well as at the workspace cap."""
    from rna_algos_amd.durbin_algo import AlignScores, durbin_algo_batch, with_pseudo_bases
    rng = np.random.default_rng(5)
    seqs = [with_pseudo_bases(rng.integers(0, 4, int(n))) for n in rng.integers(0, 3, 380)]
    pairs = [(a, b) for a in range(len(seqs)) for b in range(a + 1, len(seqs))]
    assert len(pairs) > 65535 + 4000
    sc = AlignScores.new(0.0)
    sc.transfer()
    mats = durbin_algo_batch(seqs, pairs, sc)
    assert len(mats) == len(pairs)
    for x in list(range(0, len(pairs), 997)) + [65534, 65535, 65536, len(pairs) - 1]:
        a, b = pairs[x]
        same_bits(mats[x], O.durbin(sc.ptr, seqs[a], seqs[b]), f"pair {x} = ({a},{b})")
