"""N-version guard for the index order of the loop-score tables.

oracle/bruteforce.c scores structures through the same oracle_scoring.h as the DP restatement,
so a mis-indexed table in that header would be invisible to every other test.  This file is a
SECOND restatement of the reference's 2-loop, multibranch-close and accessible scores
(/root/reference/src/utils.rs:207-366, 368-411, 423-556; src/mccaskill_algo.rs:437-455), written
from the Rust source alone in plain Python (it shares no code with oracle_scoring.h or with
rna_algos_amd/csrc/rnamc_scoring.h), compared bit for bit with what the oracle's inside pass
records (rnamc_oracle_fold_scores: every (i,j,k,l) the reference inserts into twoloop_scores)
under INDEX-ENCODING tables: entry x of table t holds t + x * 2^-16, so a swapped or transposed
index changes the value.  f32 arithmetic in the source's own association order.
"""
import numpy as np
import pytest

import oracle_lib as O

F = np.float32
A, C_, G, U = 0, 1, 2, 3
MAX_INTERIOR_EXPLICIT = 4


def augu(p):
    return p in ((A, U), (U, A), (G, U), (U, G))  # utils.rs:558-560


@pytest.fixture(scope="module")
def enc(built):
    """index-encoding parameter block: table t, flat entry x -> t + x / 65536"""
    from rna_algos_amd.utils import FoldScoreSets
    P = FoldScoreSets.new(0.0)
    names = sorted(P._fields)
    for t, name in enumerate(names, start=1):
        off, cnt = P._fields[name]
        arr = P._buf[off:off + 4 * cnt].view(np.float32)
        arr[:] = (F(t) + np.arange(cnt, dtype=np.float64) / 65536.0).astype(np.float32)
    # scalars with a meaning: both branches of max(NINIO_COEFF * diff, NINIO_MAX) get taken
    P.turner("ninio_coeff")[...] = F(-1.25)
    P.turner("ninio_max")[...] = F(-9.0)
    return P


def T(P, name):
    return P.turner(name)


# ---- Turner, utils.rs:207-366 ---------------------------------------------------------------
def turner_2loop(P, s, i, j, k, l):
    if i + 1 == k and j - 1 == l:                      # :212-215 stack
        return T(P, "stack_scores")[s[i], s[j], s[k], s[l]]
    if i + 1 == k or j - 1 == l:                       # :216-219 bulge
        ln = k - i + j - l - 2
        if ln == 1:                                    # :242-243
            return T(P, "bulge_scores_init")[1] + T(P, "stack_scores")[s[i], s[j], s[k], s[l]]
        r = T(P, "bulge_scores_init")[ln]              # :246-257
        r = r + (T(P, "helix_augu_end_penalty")[0] if augu((s[i], s[j])) else F(0))
        r = r + (T(P, "helix_augu_end_penalty")[0] if augu((s[k], s[l])) else F(0))
        return r
    a, b = k - i - 1, j - l - 1                        # :267-270
    bc, ba = (s[i], s[j]), (s[k], s[l])
    if (a, b) == (1, 1):                               # :273-277
        return T(P, "interior_scores_1x1")[bc[0], bc[1], s[i + 1], s[j - 1], ba[0], ba[1]]
    if (a, b) == (1, 2):                               # :278-285
        return T(P, "interior_scores_1x2")[bc[0], bc[1], s[i + 1], s[j - 1], s[j - 2], ba[0], ba[1]]
    if (a, b) == (2, 1):                               # :286-297: inverted pairs
        return T(P, "interior_scores_1x2")[ba[1], ba[0], s[j - 1], s[i + 2], s[i + 1], bc[1], bc[0]]
    if (a, b) == (2, 2):                               # :298-305
        return T(P, "interior_scores_2x2")[bc[0], bc[1], s[i + 1], s[j - 1], s[i + 2], s[j - 2], ba[0], ba[1]]
    # :306-319
    nin = T(P, "ninio_coeff")[0] * F(abs(a - b))
    r = T(P, "interior_scores_init")[a + b] + max(nin, T(P, "ninio_max")[0])
    # get_interior_mismatch_score :331-366 — the accessible pair is taken as (seq[l], seq[k])
    tm0, tm1 = (s[i + 1], s[j - 1]), (s[l + 1], s[k - 1])
    if a == 1 or b == 1:
        tab = T(P, "terminal_mismatch_scores_1xmany")
    elif (a, b) in ((2, 3), (3, 2)):
        tab = T(P, "terminal_mismatch_scores_2x3")
    else:
        tab = T(P, "terminal_mismatch_scores_interior")
    r = r + (tab[bc[0], bc[1], tm0[0], tm0[1]] + tab[s[l], s[k], tm1[0], tm1[1]])
    r = r + (T(P, "helix_augu_end_penalty")[0] if augu(bc) else F(0))
    r = r + (T(P, "helix_augu_end_penalty")[0] if augu(ba) else F(0))
    return r


def turner_mbclose(P, s, i, j):                        # :368-382
    tm = T(P, "terminal_mismatch_scores_multibranch")[s[j], s[i], s[j - 1], s[i + 1]]
    r = T(P, "init_multibranch_base")[0] + tm
    return r + (T(P, "helix_augu_end_penalty")[0] if augu((s[i], s[j])) else F(0))


def turner_accessible(P, s, i, j):                     # :384-411, uses_sentinel_bases = false
    n = len(s)
    if i > 0 and j < n - 1:
        r = T(P, "terminal_mismatch_scores_multibranch")[s[i], s[j], s[i - 1], s[j + 1]]
    elif i > 0:
        r = T(P, "dangling_scores_5prime")[s[i], s[j], s[i - 1]]
    elif j < n - 1:
        r = T(P, "dangling_scores_3prime")[s[i], s[j], s[j + 1]]
    else:
        r = F(0)
    return r + (T(P, "helix_augu_end_penalty")[0] if augu((s[i], s[j])) else F(0))


# ---- CONTRAfold, utils.rs:423-556 ------------------------------------------------------------
def junction_single(P, s, y0, y1):                     # :545-556
    return P.helix_close_scores[s[y0], s[y1]] + P.terminal_mismatch_scores[s[y0], s[y1], s[y0 + 1], s[y1 - 1]]


def contra_2loop(P, s, i, j, k, l):                    # :423-442
    if i + 1 == k and j - 1 == l:
        sc = P.stack_scores[s[i], s[j], s[k], s[l]]    # :444-454
    elif i + 1 == k or j - 1 == l:                     # :456-481
        ln = k - i + j - l - 2
        sc = P.bulge_scores_0x1[s[i + 1] if k - i - 1 == 1 else s[j - 1]] if ln == 1 else F(0)
        sc = sc + P.bulge_scores_len_cumulative[ln - 1]
        sc = sc + junction_single(P, s, i, j)
        sc = sc + junction_single(P, s, l, k)
    else:                                              # :483-520
        a, b = k - i - 1, j - l - 1
        if a == b:
            s11 = P.interior_scores_1x1[s[i + 1], s[j - 1]] if a + b == 2 else F(0)
            sc = s11 + P.interior_scores_symmetric_cumulative[a - 1]
        else:
            sc = P.interior_scores_asymmetric_cumulative[abs(a - b) - 1]
        se = P.interior_scores_explicit[a - 1, b - 1] if (a <= MAX_INTERIOR_EXPLICIT and b <= MAX_INTERIOR_EXPLICIT) else F(0)
        sc = sc + se
        sc = sc + P.interior_scores_len_cumulative[a + b - 2]
        sc = sc + junction_single(P, s, i, j)
        sc = sc + junction_single(P, s, l, k)
    return sc + P.basepair_scores[s[k], s[l]]          # :441


def junction(P, s, p0, p1):                            # :522-543, uses_sentinel_bases = false
    n = len(s)
    r = P.helix_close_scores[s[p0], s[p1]]
    r = r + (P.dangling_scores_left[s[p0], s[p1], s[p0 + 1]] if p0 < n - 1 else F(0))
    r = r + (P.dangling_scores_right[s[p0], s[p1], s[p1 - 1]] if p1 > 0 else F(0))
    return r


def contra_mbclose(P, s, i, j):                        # mccaskill_algo.rs:437-444
    r = P.multibranch_score_base[0] + P.multibranch_score_basepair[0]
    return r + junction(P, s, i, j)


def contra_accessible(P, s, i, j):                     # mccaskill_algo.rs:449-455
    return junction(P, s, j, i) + P.basepair_scores[s[i], s[j]]


def tri(n, i, j):
    d = j - i
    return d * n - d * (d - 1) // 2 + i


@pytest.mark.parametrize("contra,short", [(False, False), (True, False), (True, True)])
def test_second_restatement_agrees_with_oracle_scoring(enc, contra, short):
    rng = np.random.default_rng(20 + int(contra) + 2 * int(short))
    seen = set()
    with np.errstate(over="ignore", invalid="ignore"):
        for n in (48, 41, 37):
            # GC / AU / GU rich: many pairs of every type, every small-loop class occurs
            s = rng.integers(0, 4, n).astype(np.uint8)
            hp, mb, ac, tl = O.fold_scores(enc.ptr, s, contra, short)
            assert len(tl) > 2000
            two = contra_2loop if contra else turner_2loop
            for e in tl:
                i, j, k, l = int(e["i"]), int(e["j"]), int(e["k"]), int(e["l"])
                want = F(two(enc, s, i, j, k, l))
                assert want == e["score"], (contra, (i, j, k, l), (k - i - 1, j - l - 1), want, e["score"])
                a, b = k - i - 1, j - l - 1
                seen.add((min(a, 4), min(b, 4)))
            mbf = contra_mbclose if contra else turner_mbclose
            acf = contra_accessible if contra else turner_accessible
            for d in range(1, n):
                for i in range(n - d):
                    x = tri(n, i, i + d)
                    if not np.isnan(mb[x]):
                        assert F(mbf(enc, s, i, i + d)) == mb[x], ("mbclose", i, i + d)
                        assert F(acf(enc, s, i, i + d)) == ac[x], ("accessible", i, i + d)
    # every loop class was exercised: stack, both bulge sides, 1x1, 1x2, 2x1, 2x2, 2x3, 3x2, 1xn, nx1, generic
    for need in [(0, 0), (0, 1), (1, 0), (0, 3), (3, 0), (1, 1), (1, 2), (2, 1), (2, 2), (2, 3), (3, 2),
                 (1, 4), (4, 1), (4, 4), (3, 3)]:
        assert need in seen, need
