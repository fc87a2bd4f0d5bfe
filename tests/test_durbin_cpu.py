"""CPU suite for SURVEY.md 8f-4 (Durbin pair-HMM): the oracle restatement of
src/durbin_algo.rs against what the reference's own test asserts and against
model-independent properties; the constants against the reference's generated file."""
import itertools
import os
import re

import numpy as np

import oracle_lib as O

REF = "/root/reference/src/compiled_align_scores.rs"


def scores(init=None):
    from rna_algos_amd.durbin_algo import AlignScores
    s = AlignScores.new(0.0 if init is None else init)
    if init is None:
        s.transfer()
    return s


def pseudo(seq):
    from rna_algos_amd.durbin_algo import with_pseudo_bases
    return with_pseudo_bases(seq)


def test_transfer_holds_the_compiled_constants():
    """AlignScores::transfer (src/durbin_algo.rs:42-57) against the committed values of
    src/compiled_align_scores.rs:2-19; compared with the reference file itself when it is
    there (it is not on the GPU box)."""
    s = scores()
    want = dict(match2match_score=2.50575671, match2insert_score=0.1970448791,
                insert_extend_score=1.014026583, insert_switch_score=-7.346968782,
                init_match_score=0.3959924457, init_insert_score=-0.3488104904)
    for k, v in want.items():
        assert np.float32(getattr(s, k)) == np.float32(v), k
    assert np.allclose(s.match_scores, s.match_scores.T) and s.match_scores[0, 0] == np.float32(0.5256508867)
    if os.path.exists(REF):
        txt = open(REF).read()
        nums = [np.float32(x) for x in re.findall(r"-?\d+\.\d+", txt)]
        assert len(nums) == 16 + 4 + 6
        assert np.array_equal(np.array(nums[:16], np.float32).reshape(4, 4), s.match_scores)
        assert np.array_equal(np.array(nums[16:20], np.float32), s.insert_scores)
        order = ["init_match_score", "init_insert_score", "match2match_score",
                 "match2insert_score", "insert_extend_score", "insert_switch_score"]
        for k, v in zip(order, nums[20:]):
            assert np.float32(getattr(s, k)) == v, k


def test_reference_range_assertion_on_the_fixture(trnas):
    """tests/tests.rs:45-80: every match probability of all 15 pairs lies in [-0.001, 1.001)."""
    s = scores()
    seqs = [pseudo(x) for _, x in trnas]
    for a, b in itertools.combinations(range(len(seqs)), 2):
        m = O.durbin(s.ptr, seqs[a], seqs[b])
        assert m.shape == (len(seqs[a]), len(seqs[b]))
        assert m.min() >= -0.001 and m.max() < 1.001
        # borders are untouched zeros (217-229)
        assert not m[0].any() and not m[-1].any() and not m[:, 0].any() and not m[:, -1].any()


def test_posterior_properties():
    """Model-independent checks of the restatement: a nucleotide matches at most one partner
    (row / column sums <= 1 up to the cubic logsumexp / expf error), identical sequences put
    their mass on the diagonal, and with all-zero scores every alignment path weighs the same,
    so the match probability is a ratio of Delannoy-type path counts."""
    rng = np.random.default_rng(3)
    s = scores()
    for n1, n2 in ((5, 9), (30, 30), (64, 41), (2, 7), (1, 1)):
        a, b = pseudo(rng.integers(0, 4, n1)), pseudo(rng.integers(0, 4, n2))
        m = O.durbin(s.ptr, a, b).astype(np.float64)
        assert m.sum(axis=1).max() <= 1.0 + 5e-3 and m.sum(axis=0).max() <= 1.0 + 5e-3
    a = pseudo(rng.integers(0, 4, 40))
    m = O.durbin(s.ptr, a, a)
    assert np.all(np.argmax(m[1:-1, 1:-1], axis=1) == np.arange(40))
    # zero scores: P(i ~ j) = F(i, j) * B(i, j) / total with F, B path counts of the 3-state HMM
    z = scores(0.0)
    n1, n2 = 6, 5
    a, b = pseudo(rng.integers(0, 4, n1)), pseudo(rng.integers(0, 4, n2))
    m = O.durbin(z.ptr, a, b).astype(np.float64)
    N1, N2 = n1 + 2, n2 + 2

    def counts(forward):
        M = np.zeros((N1, N2)); I = np.zeros((N1, N2)); D = np.zeros((N1, N2))
        rng_i = range(0, N1 - 1) if forward else range(N1 - 1, 0, -1)
        rng_j = range(0, N2 - 1) if forward else range(N2 - 1, 0, -1)
        st = 1 if forward else -1
        for i in rng_i:
            for j in rng_j:
                if (i, j) == ((0, 0) if forward else (N1 - 1, N2 - 1)):
                    M[i, j] = 1
                    continue
                pi, pj = i - st, j - st
                inside = (i > 0 and j > 0) if forward else (i < N1 - 1 and j < N2 - 1)
                if inside:
                    M[i, j] = M[pi, pj] + I[pi, pj] + D[pi, pj]
                if (i > 0) if forward else (i < N1 - 1):
                    I[i, j] = M[pi, j] + I[pi, j]
                if (j > 0) if forward else (j < N2 - 1):
                    D[i, j] = M[i, pj] + D[i, pj]
        return M, I, D
    FM, FI, FD = counts(True)
    BM, BI, BD = counts(False)
    total = FM[N1 - 2, N2 - 2] + FI[N1 - 2, N2 - 2] + FD[N1 - 2, N2 - 2]
    for i in range(1, N1 - 1):
        for j in range(1, N2 - 1):
            want = FM[i, j] * (BM[i + 1, j + 1] + BI[i + 1, j + 1] + BD[i + 1, j + 1]) / total
            assert abs(m[i, j] - want) < 2e-3, (i, j, m[i, j], want)
