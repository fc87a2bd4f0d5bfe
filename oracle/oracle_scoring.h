/*
 * oracle_scoring.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement, in plain C, of the scoring and numeric helpers of the
 * reference's McCaskill path.  Every function names the reference lines it
 * follows (paths relative to /root/reference).  Only tests/, the smoke check
 * and bench.py's cpu_baseline leg may use anything under oracle/; the product
 * (rna_algos_amd/) never includes, links or calls it.
 *
 * PARITY UNPINNED: the reference cannot be built here (no Rust toolchain) and
 * its numeric tables live in the absent crate rna-ss-params 0.1, so this
 * restatement is pinned only by (a) the reference's own range assertion
 * (tests/tests.rs:33,38), (b) a brute-force structure enumeration in f64
 * (oracle/bruteforce.c) and (c) all-zero-table structure counts.
 *
 * Arithmetic rules: f32 everywhere, expression association exactly as the Rust
 * source parses (left to right), no FMA contraction (build with
 * -ffp-contract=off), no fast-math.
 */
#ifndef ORACLE_SCORING_H
#define ORACLE_SCORING_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../include/rnamc.h"

#ifdef ORACLE_EXACT
/* NOT the reference's arithmetic: the same recurrences and loop scores evaluated in f64
 * with an exact logsumexp / exp (mccaskill_exact.c).  It checks the tree-order summation
 * mode of the HIP path (rnamc_ctx_set "summation_mode" 1), which cannot be bit-compared
 * with the reference's order-dependent fold. */
typedef double Score;
#else
typedef float Score;
#endif
#define ONEG_INF (-INFINITY)

/* src/utils.rs:162-164 — AU | CG | GC | GU | UA | UG with A,C,G,U = 0,1,2,3. */
static inline int o_has_canonical_basepair(int a, int b) {
  return (a == 0 && b == 3) || (a == 1 && b == 2) || (a == 2 && b == 1) || (a == 2 && b == 3) ||
         (a == 3 && b == 0) || (a == 3 && b == 2);
}

/* src/utils.rs:558-560 */
static inline int o_matches_augu(int a, int b) {
  return (a == 0 && b == 3) || (a == 3 && b == 0) || (a == 2 && b == 3) || (a == 3 && b == 2);
}

/* src/utils.rs:602-627 */
static inline Score o_ln_exp_1p(Score x) {
  if (x < 3.3792500f) {
    if (x < 1.6320158f) {
      if (x < 0.66153675f) {
        return ((-0.0065591595f * x + 0.12764427f) * x + 0.49965546f) * x + 0.6931542f;
      } else {
        return ((-0.015515756f * x + 0.14467756f) * x + 0.48829398f) * x + 0.6958093f;
      }
    } else if (x < 2.4912589f) {
      return ((-0.012890925f * x + 0.13010283f) * x + 0.51503986f) * x + 0.6795586f;
    } else {
      return ((-0.0072142647f * x + 0.087754086f) * x + 0.6208708f) * x + 0.5909676f;
    }
  } else if (x < 5.789071f) {
    if (x < 4.426169f) {
      return ((-0.0031455354f * x + 0.046722945f) * x + 0.7592532f) * x + 0.43487945f;
    } else {
      return ((-0.0010110698f * x + 0.018594341f) * x + 0.88317305f) * x + 0.25236955f;
    }
  } else if (x < 7.8162727f) {
    return ((-0.000196278f * x + 0.0046084408f) * x + 0.9634432f) * x + 0.09831489f;
  } else {
    return ((-0.0000113994f * x + 0.0003734731f) * x + 0.9959107f) * x + 0.0149855051f;
  }
}

/* src/utils.rs:579-596 (LOGSUMEXP_THRESHOLD_UPPER at src/utils.rs:121) */
static inline void o_logsumexp(Score* sum, Score x) {
  if (!isfinite(x)) return;
  if (!isfinite(*sum)) {
    *sum = x;
  } else {
#ifdef ORACLE_EXACT
    Score hi = *sum > x ? *sum : x, lo = *sum > x ? x : *sum;
    *sum = hi + log1p(exp(lo - hi));
#else
    Score y = fminf(*sum, x);
    Score z = fmaxf(*sum, x) - y;
    *sum = y + (z >= 11.862479f ? z : o_ln_exp_1p(z));
#endif
  }
}

/* src/utils.rs:630-655 */
static inline Score o_expf(Score x) {
#ifdef ORACLE_EXACT
  return exp(x);
#else
  if (x < -2.4915035f) {
    if (x < -5.8622823f) {
      if (x < -9.91152f) {
        return 0.f;
      } else {
        return ((0.0000803850f * x + 0.002162743f) * x + 0.019470856f) * x + 0.058808003f;
      }
    } else if (x < -3.839663f) {
      return ((0.0013889414f * x + 0.024467647f) * x + 0.14712906f) * x + 0.30427578f;
    } else {
      return ((0.0072335607f * x + 0.09060027f) * x + 0.39831114f) * x + 0.62459594f;
    }
  } else if (x < -0.6725053f) {
    if (x < -1.4805375f) {
      return ((0.023241036f * x + 0.2085646f) * x + 0.6906368f) * x + 0.86823225f;
    } else {
      return ((0.057378277f * x + 0.35802585f) * x + 0.9121133f) * x + 0.9793092f;
    }
  } else if (x < 0.f) {
    return ((0.119917594f * x + 0.48156682f) * x + 0.9975992f) * x + 0.9999505f;
  } else {
    return expf(x);
  }
#endif
}

/* ------------------------- Turner model, src/utils.rs:166-411 ------------- */

/* src/utils.rs:198-205 */
static inline Score o_get_special_hairpin_score(const rnamc_turner_scores* t, const uint8_t* hp,
                                                uint32_t len) {
  for (uint32_t x = 0; x < t->num_special_hairpins; x++) {
    if (t->special_hairpin_lens[x] == len && memcmp(t->special_hairpin_seqs[x], hp, len) == 0) {
      return t->special_hairpin_scores[x];
    }
  }
  return ONEG_INF;
}

/* src/utils.rs:166-196 */
static inline Score o_get_hairpin_score(const rnamc_turner_scores* t, const uint8_t* seq,
                                        uint32_t i, uint32_t j) {
  Score special = o_get_special_hairpin_score(t, seq + i, j - i + 1);
  if (special > ONEG_INF) return special;
  uint32_t hairpin_len = j - i - 1;
  int bi = seq[i], bj = seq[j];
  Score hairpin_score;
  if (hairpin_len == t->min_hairpin_len) {
    hairpin_score = t->hairpin_scores_init[hairpin_len];
  } else {
    int t0 = seq[i + 1], t1 = seq[j - 1];
    Score init;
    if (hairpin_len <= t->max_hairpin_len_extrapolation) {
      init = t->hairpin_scores_init[hairpin_len];
    } else {
      uint32_t m = t->min_hairpin_len_extrapolation - 1;
      init = t->hairpin_scores_init[m] +
             t->coeff_hairpin_len_extrapolation * logf((Score)hairpin_len / (Score)m);
    }
    hairpin_score = init + t->terminal_mismatch_scores_hairpin[bi][bj][t0][t1];
  }
  return hairpin_score + (o_matches_augu(bi, bj) ? t->helix_augu_end_penalty : 0.f);
}

/* src/utils.rs:224-232 */
static inline Score o_get_stack_score(const rnamc_turner_scores* t, const uint8_t* seq, uint32_t i,
                                      uint32_t j, uint32_t k, uint32_t l) {
  return t->stack_scores[seq[i]][seq[j]][seq[k]][seq[l]];
}

/* src/utils.rs:234-258 */
static inline Score o_get_bulge_score(const rnamc_turner_scores* t, const uint8_t* seq, uint32_t i,
                                      uint32_t j, uint32_t k, uint32_t l) {
  uint32_t bulge_len = k - i + j - l - 2;
  if (bulge_len == 1) {
    return t->bulge_scores_init[bulge_len] + o_get_stack_score(t, seq, i, j, k, l);
  }
  return t->bulge_scores_init[bulge_len] +
         (o_matches_augu(seq[i], seq[j]) ? t->helix_augu_end_penalty : 0.f) +
         (o_matches_augu(seq[k], seq[l]) ? t->helix_augu_end_penalty : 0.f);
}

/* src/utils.rs:331-366 */
static inline Score o_get_interior_mismatch_score(const rnamc_turner_scores* t, const uint8_t* seq,
                                                  uint32_t i, uint32_t j, uint32_t k, uint32_t l,
                                                  uint32_t l0, uint32_t l1) {
  int c0 = seq[i], c1 = seq[j];
  int a0 = seq[l], a1 = seq[k]; /* basepair_accessible is (seq[l], seq[k]) here */
  int m00 = seq[i + 1], m01 = seq[j - 1];
  int m10 = seq[l + 1], m11 = seq[k - 1];
  if (l0 == 1 || l1 == 1) {
    return t->terminal_mismatch_scores_1xmany[c0][c1][m00][m01] +
           t->terminal_mismatch_scores_1xmany[a0][a1][m10][m11];
  } else if ((l0 == 2 && l1 == 3) || (l0 == 3 && l1 == 2)) {
    return t->terminal_mismatch_scores_2x3[c0][c1][m00][m01] +
           t->terminal_mismatch_scores_2x3[a0][a1][m10][m11];
  }
  return t->terminal_mismatch_scores_interior[c0][c1][m00][m01] +
         t->terminal_mismatch_scores_interior[a0][a1][m10][m11];
}

/* src/utils.rs:260-321 */
static inline Score o_get_interior_score(const rnamc_turner_scores* t, const uint8_t* seq,
                                         uint32_t i, uint32_t j, uint32_t k, uint32_t l) {
  int c0 = seq[i], c1 = seq[j];
  int a0 = seq[k], a1 = seq[l];
  uint32_t l0 = k - i - 1, l1 = j - l - 1;
  uint32_t interior_len = l0 + l1;
  if (l0 == 1 && l1 == 1) {
    return t->interior_scores_1x1[c0][c1][seq[i + 1]][seq[j - 1]][a0][a1];
  } else if (l0 == 1 && l1 == 2) {
    return t->interior_scores_1x2[c0][c1][seq[i + 1]][seq[j - 1]][seq[j - 2]][a0][a1];
  } else if (l0 == 2 && l1 == 1) {
    /* interior = ((seq[j-1], seq[i+2]), seq[i+1]); both pairs inverted */
    return t->interior_scores_1x2[a1][a0][seq[j - 1]][seq[i + 2]][seq[i + 1]][c1][c0];
  } else if (l0 == 2 && l1 == 2) {
    return t->interior_scores_2x2[c0][c1][seq[i + 1]][seq[j - 1]][seq[i + 2]][seq[j - 2]][a0][a1];
  }
  uint32_t diff = l0 > l1 ? l0 - l1 : l1 - l0;
  return t->interior_scores_init[interior_len] + fmaxf(t->ninio_coeff * (Score)diff, t->ninio_max) +
         o_get_interior_mismatch_score(t, seq, i, j, k, l, l0, l1) +
         (o_matches_augu(c0, c1) ? t->helix_augu_end_penalty : 0.f) +
         (o_matches_augu(a0, a1) ? t->helix_augu_end_penalty : 0.f);
}

/* src/utils.rs:207-222 */
static inline Score o_get_2loop_score(const rnamc_turner_scores* t, const uint8_t* seq, uint32_t i,
                                      uint32_t j, uint32_t k, uint32_t l) {
  if (i + 1 == k && j - 1 == l) return o_get_stack_score(t, seq, i, j, k, l);
  if (i + 1 == k || j - 1 == l) return o_get_bulge_score(t, seq, i, j, k, l);
  return o_get_interior_score(t, seq, i, j, k, l);
}

/* src/utils.rs:368-382 */
static inline Score o_get_multibranch_close_score(const rnamc_turner_scores* t, const uint8_t* seq,
                                                  uint32_t i, uint32_t j) {
  int c0 = seq[i], c1 = seq[j];
  /* inverses: close (c1,c0); stack (seq[j-1], seq[i+1]) */
  Score tm = t->terminal_mismatch_scores_multibranch[c1][c0][seq[j - 1]][seq[i + 1]];
  return t->init_multibranch_base + tm + (o_matches_augu(c0, c1) ? t->helix_augu_end_penalty : 0.f);
}

/* src/utils.rs:384-411 with uses_sentinel_bases = false
 * (src/mccaskill_algo.rs:287) */
static inline Score o_get_accessible_score(const rnamc_turner_scores* t, const uint8_t* seq,
                                           uint32_t n, uint32_t i, uint32_t j) {
  uint32_t end_5prime = 0, end_3prime = n - 1;
  int a0 = seq[i], a1 = seq[j];
  Score score;
  if (i > end_5prime && j < end_3prime) {
    score = t->terminal_mismatch_scores_multibranch[a0][a1][seq[i - 1]][seq[j + 1]];
  } else if (i > end_5prime) {
    score = t->dangling_scores_5prime[a0][a1][seq[i - 1]];
  } else if (j < end_3prime) {
    score = t->dangling_scores_3prime[a0][a1][seq[j + 1]];
  } else {
    score = 0.f;
  }
  return score + (o_matches_augu(a0, a1) ? t->helix_augu_end_penalty : 0.f);
}

/* ---------------------- CONTRAfold model, src/utils.rs:413-556 ------------ */

/* src/utils.rs:545-548: pair (x[y0], x[y1]), mismatch (x[y0+1], x[y1-1]) */
static inline Score o_get_junction_score_single(const rnamc_fold_score_sets* f, const uint8_t* seq,
                                                uint32_t y0, uint32_t y1) {
  int a0 = seq[y0], a1 = seq[y1];
  return f->helix_close_scores[a0][a1] +
         f->terminal_mismatch_scores[a0][a1][seq[y0 + 1]][seq[y1 - 1]];
}

/* src/utils.rs:522-543 with uses_sentinel_bases = false */
static inline Score o_get_junction_score(const rnamc_fold_score_sets* f, const uint8_t* seq,
                                         uint32_t n, uint32_t p0, uint32_t p1) {
  int b0 = seq[p0], b1 = seq[p1];
  uint32_t end_5prime = 0, end_3prime = n - 1;
  return f->helix_close_scores[b0][b1] +
         (p0 < end_3prime ? f->dangling_scores_left[b0][b1][seq[p0 + 1]] : 0.f) +
         (p1 > end_5prime ? f->dangling_scores_right[b0][b1][seq[p1 - 1]] : 0.f);
}

/* src/utils.rs:413-421 */
static inline Score o_get_hairpin_score_contra(const rnamc_fold_score_sets* f, const uint8_t* seq,
                                               uint32_t i, uint32_t j) {
  uint32_t hairpin_len = j - i - 1;
  uint32_t idx = hairpin_len < RNAMC_MAX_LOOP_LEN ? hairpin_len : RNAMC_MAX_LOOP_LEN;
  return f->hairpin_scores_len_cumulative[idx] + o_get_junction_score_single(f, seq, i, j);
}

/* src/utils.rs:444-454 */
static inline Score o_get_stack_score_contra(const rnamc_fold_score_sets* f, const uint8_t* seq,
                                             uint32_t i, uint32_t j, uint32_t k, uint32_t l) {
  return f->stack_scores[seq[i]][seq[j]][seq[k]][seq[l]];
}

/* src/utils.rs:456-481 */
static inline Score o_get_bulge_score_contra(const rnamc_fold_score_sets* f, const uint8_t* seq,
                                             uint32_t i, uint32_t j, uint32_t k, uint32_t l) {
  uint32_t bulge_len = k - i + j - l - 2;
  Score score = 0.f;
  if (bulge_len == 1) {
    score = f->bulge_scores_0x1[(k - i - 1 == 1) ? seq[i + 1] : seq[j - 1]];
  }
  return score + f->bulge_scores_len_cumulative[bulge_len - 1] +
         o_get_junction_score_single(f, seq, i, j) + o_get_junction_score_single(f, seq, l, k);
}

/* src/utils.rs:483-520 */
static inline Score o_get_interior_score_contra(const rnamc_fold_score_sets* f, const uint8_t* seq,
                                                uint32_t i, uint32_t j, uint32_t k, uint32_t l) {
  uint32_t l0 = k - i - 1, l1 = j - l - 1;
  uint32_t interior_len = l0 + l1;
  Score score;
  if (l0 == l1) {
    Score score_1x1 = (interior_len == 2) ? f->interior_scores_1x1[seq[i + 1]][seq[j - 1]] : 0.f;
    score = score_1x1 + f->interior_scores_symmetric_cumulative[l0 - 1];
  } else {
    uint32_t diff = l0 > l1 ? l0 - l1 : l1 - l0;
    score = f->interior_scores_asymmetric_cumulative[diff - 1];
  }
  Score score_explicit = (l0 <= RNAMC_MAX_INTERIOR_EXPLICIT && l1 <= RNAMC_MAX_INTERIOR_EXPLICIT)
                             ? f->interior_scores_explicit[l0 - 1][l1 - 1]
                             : 0.f;
  return score + score_explicit + f->interior_scores_len_cumulative[interior_len - 2] +
         o_get_junction_score_single(f, seq, i, j) + o_get_junction_score_single(f, seq, l, k);
}

/* src/utils.rs:423-442 */
static inline Score o_get_2loop_score_contra(const rnamc_fold_score_sets* f, const uint8_t* seq,
                                             uint32_t i, uint32_t j, uint32_t k, uint32_t l) {
  Score score;
  if (i + 1 == k && j - 1 == l) {
    score = o_get_stack_score_contra(f, seq, i, j, k, l);
  } else if (i + 1 == k || j - 1 == l) {
    score = o_get_bulge_score_contra(f, seq, i, j, k, l);
  } else {
    score = o_get_interior_score_contra(f, seq, i, j, k, l);
  }
  return score + f->basepair_scores[seq[k]][seq[l]];
}

/* multibranch close score of the CONTRAfold branch, src/mccaskill_algo.rs:437-444 */
static inline Score o_get_multibranch_close_score_contra(const rnamc_fold_score_sets* f,
                                                         const uint8_t* seq, uint32_t n, uint32_t i,
                                                         uint32_t j) {
  return f->multibranch_score_base + f->multibranch_score_basepair +
         o_get_junction_score(f, seq, n, i, j);
}

/* accessible score of the CONTRAfold branch, src/mccaskill_algo.rs:449-455 */
static inline Score o_get_accessible_score_contra(const rnamc_fold_score_sets* f,
                                                  const uint8_t* seq, uint32_t n, uint32_t i,
                                                  uint32_t j) {
  return o_get_junction_score(f, seq, n, j, i) + f->basepair_scores[seq[i]][seq[j]];
}

#endif /* ORACLE_SCORING_H */
