/*
 * mccaskill_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Loop-for-loop CPU restatement of the reference's McCaskill inside/outside DP
 * (/root/reference/src/mccaskill_algo.rs:213-723), used as the parity oracle for
 * the HIP path and as the timed CPU baseline ("port") in bench.py.
 * PARITY UNPINNED — see oracle_scoring.h.
 *
 * Differences from the Rust source, all value-preserving:
 *  - hashbrown maps (sums_close, sums_accessible, basepair_probs and the
 *    FoldScores maps) are dense n*n f32 arrays holding -inf for "no entry":
 *    only finite values are ever inserted (mccaskill_algo.rs:332,456,602,715)
 *    and logsumexp ignores non-finite terms (utils.rs:581);
 *  - twoloop_scores is not materialised: the outside pass recomputes the same
 *    pure function of (seq, k, l, i, j) that the inside pass evaluated
 *    (mccaskill_algo.rs:319-320 / 589).
 */
#include <pthread.h>
#include <stdlib.h>

#include "oracle_scoring.h"

#define IDX(i, j) ((size_t)(i) * n + (size_t)(j))

typedef struct {
  uint32_t n;
  /* FoldSums, mccaskill_algo.rs:3-11, init at 213-226 */
  Score* sums_external;                        /* 0.0 everywhere        */
  Score* sums_rightmost_basepairs_external;    /* -inf                  */
  Score* sums_rightmost_basepairs_multibranch; /* -inf                  */
  Score* sums_close;                           /* sparse -> -inf        */
  Score* sums_accessible;                      /* sparse -> -inf        */
  Score* sums_multibranch;                     /* -inf                  */
  Score* sums_1ormore_basepairs;               /* -inf                  */
  /* FoldScores (dense), mccaskill_algo.rs:13-19 */
  Score* multibranch_close_scores;
  /* outside, mccaskill_algo.rs:527-529 */
  Score* basepair_probs; /* log domain until the final expf map; -inf = absent */
  Score* probs_multibranch;
  Score* probs_multibranch2;
  /* optional recorder of the remaining FoldScores maps (rnamc_oracle_fold_scores):
   * dense n x n with NaN = key absent, and the list of twoloop_scores inserts */
  Score* rec_hairpin;
  Score* rec_accessible;
  rnamc_twoloop_score* rec_twoloop;
  size_t rec_count, rec_cap;
  int rec_failed;
} ostate;

static void rec_twoloop(ostate* s, uint32_t i, uint32_t j, uint32_t k, uint32_t l, float y) {
  if (!s->rec_hairpin) return;
  if (s->rec_count == s->rec_cap) {
    size_t cap = s->rec_cap ? s->rec_cap * 2 : 4096;
    rnamc_twoloop_score* q = (rnamc_twoloop_score*)realloc(s->rec_twoloop, cap * sizeof(*q));
    if (!q) {
      s->rec_failed = 1;
      return;
    }
    s->rec_twoloop = q;
    s->rec_cap = cap;
  }
  rnamc_twoloop_score e = {i, j, k, l, y};
  s->rec_twoloop[s->rec_count++] = e;
}

static Score* alloc_fill(size_t count, Score v) {
  Score* p = (Score*)malloc(count * sizeof(Score));
  if (!p) return NULL;
  for (size_t x = 0; x < count; x++) p[x] = v;
  return p;
}

static void ostate_free(ostate* s) {
  free(s->sums_external);
  free(s->sums_rightmost_basepairs_external);
  free(s->sums_rightmost_basepairs_multibranch);
  free(s->sums_close);
  free(s->sums_accessible);
  free(s->sums_multibranch);
  free(s->sums_1ormore_basepairs);
  free(s->multibranch_close_scores);
  free(s->basepair_probs);
  free(s->probs_multibranch);
  free(s->probs_multibranch2);
  free(s->rec_hairpin);
  free(s->rec_accessible);
  free(s->rec_twoloop);
  memset(s, 0, sizeof(*s));
}

static int ostate_init(ostate* s, uint32_t n) {
  size_t c = (size_t)n * n;
  memset(s, 0, sizeof(*s));
  s->n = n;
  s->sums_external = alloc_fill(c, 0.f);
  s->sums_rightmost_basepairs_external = alloc_fill(c, ONEG_INF);
  s->sums_rightmost_basepairs_multibranch = alloc_fill(c, ONEG_INF);
  s->sums_close = alloc_fill(c, ONEG_INF);
  s->sums_accessible = alloc_fill(c, ONEG_INF);
  s->sums_multibranch = alloc_fill(c, ONEG_INF);
  s->sums_1ormore_basepairs = alloc_fill(c, ONEG_INF);
  s->multibranch_close_scores = alloc_fill(c, ONEG_INF);
  s->basepair_probs = alloc_fill(c, ONEG_INF);
  s->probs_multibranch = alloc_fill(c, ONEG_INF);
  s->probs_multibranch2 = alloc_fill(c, ONEG_INF);
  if (!s->sums_external || !s->sums_rightmost_basepairs_external ||
      !s->sums_rightmost_basepairs_multibranch || !s->sums_close || !s->sums_accessible ||
      !s->sums_multibranch || !s->sums_1ormore_basepairs || !s->multibranch_close_scores ||
      !s->basepair_probs || !s->probs_multibranch || !s->probs_multibranch2) {
    ostate_free(s);
    return RNAMC_ERR_OOM;
  }
  return RNAMC_OK;
}

/* get_fold_sums — mccaskill_algo.rs:282-378 */
static void o_get_fold_sums(const rnamc_params* p, const uint8_t* seq, ostate* s) {
  const rnamc_turner_scores* t = &p->turner;
  const uint32_t n = s->n;
  for (uint32_t subseq_len = RNAMC_MIN_SPAN_HAIRPIN_CLOSE; subseq_len <= n; subseq_len++) {
    for (uint32_t i = 0; i + subseq_len <= n; i++) {
      uint32_t j = i + subseq_len - 1;
      Score sum = ONEG_INF;
      if (j - i + 1 >= RNAMC_MIN_SPAN_HAIRPIN_CLOSE && o_has_canonical_basepair(seq[i], seq[j])) {
        Score hairpin_score = o_get_hairpin_score(t, seq, i, j);
        if (s->rec_hairpin) s->rec_hairpin[IDX(i, j)] = hairpin_score; /* :302-304 */
        o_logsumexp(&sum, hairpin_score);
        for (uint32_t k = i + 1; k < j - 1; k++) { /* range(i+1, j-1) */
          if (k - i - 1 > RNAMC_MAX_2LOOP_LEN) break;
          for (uint32_t l = j - 1; l > k; l--) { /* range(k+1, j).rev() */
            if (j - l - 1 + k - i - 1 > RNAMC_MAX_2LOOP_LEN) break;
            Score x = s->sums_close[IDX(k, l)];
            if (x > ONEG_INF) { /* map hit */
              Score y = o_get_2loop_score(t, seq, i, j, k, l);
              rec_twoloop(s, i, j, k, l, y); /* :320 */
              y = x + y;
              o_logsumexp(&sum, y);
            }
          }
        }
        Score multibranch_close_score = o_get_multibranch_close_score(t, seq, i, j);
        o_logsumexp(&sum, s->sums_multibranch[IDX(i + 1, j - 1)] + multibranch_close_score);
        Score accessible_score = o_get_accessible_score(t, seq, n, i, j);
        if (sum > ONEG_INF) {
          s->multibranch_close_scores[IDX(i, j)] = multibranch_close_score;
          if (s->rec_accessible) s->rec_accessible[IDX(i, j)] = accessible_score; /* :336-338 */
          s->sums_close[IDX(i, j)] = sum;
          s->sums_accessible[IDX(i, j)] = sum + accessible_score;
        }
      }
      sum = ONEG_INF;
      for (uint32_t k = i + 1; k <= j; k++) {
        Score x = s->sums_accessible[IDX(i, k)];
        if (x > ONEG_INF) o_logsumexp(&sum, x);
      }
      s->sums_rightmost_basepairs_external[IDX(i, j)] = sum;
      sum = 0.f;
      for (uint32_t k = i; k < j; k++) {
        Score x = s->sums_rightmost_basepairs_external[IDX(k, j)];
        Score y = (i == 0 && k == 0) ? 0.f : s->sums_external[IDX(i, k - 1)];
        y = x + y;
        o_logsumexp(&sum, y);
      }
      s->sums_external[IDX(i, j)] = sum;
      sum = s->sums_rightmost_basepairs_external[IDX(i, j)] + t->coeff_num_branches;
      Score sum2 = ONEG_INF;
      for (uint32_t k = i + 1; k < j; k++) {
        Score x = s->sums_rightmost_basepairs_external[IDX(k, j)] + t->coeff_num_branches;
        o_logsumexp(&sum, x);
        Score y = s->sums_1ormore_basepairs[IDX(i, k - 1)] + x;
        o_logsumexp(&sum2, y);
      }
      s->sums_multibranch[IDX(i, j)] = sum2;
      o_logsumexp(&sum, sum2);
      s->sums_1ormore_basepairs[IDX(i, j)] = sum;
    }
  }
}

/* get_fold_sums_contra — mccaskill_algo.rs:380-516 */
static void o_get_fold_sums_contra(const rnamc_params* p, const uint8_t* seq, ostate* s,
                                   int allows_short_hairpins) {
  const rnamc_fold_score_sets* f = &p->contra;
  const uint32_t n = s->n;
  for (uint32_t subseq_len = 1; subseq_len <= n; subseq_len++) {
    for (uint32_t i = 0; i + subseq_len <= n; i++) {
      uint32_t j = i + subseq_len - 1;
      Score sum = ONEG_INF;
      if (o_has_canonical_basepair(seq[i], seq[j]) &&
          (allows_short_hairpins || j - i + 1 >= RNAMC_MIN_SPAN_HAIRPIN_CLOSE)) {
        if (j - i - 1 <= RNAMC_MAX_LOOP_LEN) {
          Score hairpin_score = o_get_hairpin_score_contra(f, seq, i, j);
          if (s->rec_hairpin) s->rec_hairpin[IDX(i, j)] = hairpin_score; /* :407-409 */
          o_logsumexp(&sum, hairpin_score);
        }
        /* range(i+1, j-1): empty when j < i+2; j >= i+1 here (canonical => i != j) */
        for (uint32_t k = i + 1; k + 1 < j; k++) {
          if (k - i - 1 > RNAMC_MAX_LOOP_LEN) break;
          for (uint32_t l = j - 1; l > k; l--) {
            if (j - l - 1 + k - i - 1 > RNAMC_MAX_LOOP_LEN) break;
            Score x = s->sums_close[IDX(k, l)];
            if (x > ONEG_INF) {
              Score y = o_get_2loop_score_contra(f, seq, i, j, k, l);
              rec_twoloop(s, i, j, k, l, y); /* :431 */
              y = x + y;
              o_logsumexp(&sum, y);
            }
          }
        }
        Score multibranch_close_score = o_get_multibranch_close_score_contra(f, seq, n, i, j);
        o_logsumexp(&sum, s->sums_multibranch[IDX(i + 1, j - 1)] + multibranch_close_score);
        Score accessible_score = o_get_accessible_score_contra(f, seq, n, i, j);
        if (sum > ONEG_INF) {
          s->multibranch_close_scores[IDX(i, j)] = multibranch_close_score;
          if (s->rec_accessible) s->rec_accessible[IDX(i, j)] = accessible_score; /* :460-462 */
          s->sums_close[IDX(i, j)] = sum;
          s->sums_accessible[IDX(i, j)] = sum + accessible_score;
        }
      }
      sum = ONEG_INF;
      Score sum2 = sum;
      for (uint32_t k = i + 1; k <= j; k++) {
        Score x = s->sums_accessible[IDX(i, k)];
        if (x > ONEG_INF) {
          o_logsumexp(&sum, x + f->external_score_basepair + f->external_score_unpair * (Score)(j - k));
          o_logsumexp(&sum2,
                      x + f->multibranch_score_basepair + f->multibranch_score_unpair * (Score)(j - k));
        }
      }
      s->sums_rightmost_basepairs_external[IDX(i, j)] = sum;
      s->sums_rightmost_basepairs_multibranch[IDX(i, j)] = sum2;
      sum = f->external_score_unpair * (Score)subseq_len;
      for (uint32_t k = i; k < j; k++) {
        Score x = s->sums_rightmost_basepairs_external[IDX(k, j)];
        Score y = (i == 0 && k == 0) ? 0.f : s->sums_external[IDX(i, k - 1)];
        y = x + y;
        o_logsumexp(&sum, y);
      }
      s->sums_external[IDX(i, j)] = sum;
      sum = s->sums_rightmost_basepairs_multibranch[IDX(i, j)];
      sum2 = ONEG_INF;
      for (uint32_t k = i + 1; k < j; k++) {
        Score x = s->sums_rightmost_basepairs_multibranch[IDX(k, j)];
        o_logsumexp(&sum, x + f->multibranch_score_unpair * (Score)(k - i));
        x = s->sums_1ormore_basepairs[IDX(i, k - 1)] + x;
        o_logsumexp(&sum2, x);
      }
      s->sums_multibranch[IDX(i, j)] = sum2;
      o_logsumexp(&sum, sum2);
      s->sums_1ormore_basepairs[IDX(i, j)] = sum;
    }
  }
}

/* get_basepair_probs — mccaskill_algo.rs:518-610 (without the final expf map).
 * Returns non-zero where the reference would panic on a missing map key. */
static int o_get_basepair_probs(const rnamc_params* p, const uint8_t* seq, ostate* s) {
  const rnamc_turner_scores* t = &p->turner;
  const uint32_t n = s->n;
  const Score global_sum = s->sums_external[IDX(0, n - 1)];
  for (uint32_t subseq_len = n; subseq_len >= RNAMC_MIN_SPAN_HAIRPIN_CLOSE; subseq_len--) {
    for (uint32_t i = 0; i + subseq_len <= n; i++) {
      uint32_t j = i + subseq_len - 1;
      Score sum = ONEG_INF;
      Score sum2 = sum;
      for (uint32_t k = j + 1; k < n; k++) {
        Score x = s->sums_close[IDX(i, k)];
        if (x > ONEG_INF) {
          Score basepair_prob = s->basepair_probs[IDX(i, k)];
          if (!(basepair_prob > ONEG_INF)) return 100;
          Score multibranch_close_score = s->multibranch_close_scores[IDX(i, k)];
          x = basepair_prob + multibranch_close_score - x;
          o_logsumexp(&sum, x + s->sums_1ormore_basepairs[IDX(j + 1, k - 1)]);
          o_logsumexp(&sum2, x);
        }
      }
      s->probs_multibranch[IDX(i, j)] = sum;
      s->probs_multibranch2[IDX(i, j)] = sum2;
      Score sum_close = s->sums_close[IDX(i, j)];
      if (sum_close > ONEG_INF) {
        Score sum_accessible = s->sums_accessible[IDX(i, j)];
        Score sum_pair0 = (i < 1) ? 0.f : s->sums_external[IDX(0, i - 1)];
        Score sum_pair1 = (j > n - 2) ? 0.f : s->sums_external[IDX(j + 1, n - 1)];
        sum = sum_pair0 + sum_accessible + sum_pair1 - global_sum;
        for (uint32_t kk = i; kk-- > 0;) { /* range(0, i).rev() */
          uint32_t k = kk;
          if (i - k - 1 > RNAMC_MAX_2LOOP_LEN) break;
          for (uint32_t l = j + 1; l < n; l++) {
            if (l - j - 1 + i - k - 1 > RNAMC_MAX_2LOOP_LEN) break;
            Score x = s->sums_close[IDX(k, l)];
            if (x > ONEG_INF) {
              Score bp = s->basepair_probs[IDX(k, l)];
              if (!(bp > ONEG_INF)) return 100;
              o_logsumexp(&sum, bp + sum_close - x + o_get_2loop_score(t, seq, k, l, i, j));
            }
          }
        }
        sum_accessible = sum_accessible + t->coeff_num_branches;
        for (uint32_t k = 0; k < i; k++) {
          Score x = s->sums_1ormore_basepairs[IDX(k + 1, i - 1)];
          o_logsumexp(&sum, sum_accessible + s->probs_multibranch2[IDX(k, j)] + x);
          Score y = s->probs_multibranch[IDX(k, j)];
          o_logsumexp(&sum, sum_accessible + y);
          o_logsumexp(&sum, sum_accessible + x + y);
        }
        if (sum > ONEG_INF) s->basepair_probs[IDX(i, j)] = sum;
      }
    }
  }
  return 0;
}

/* get_basepair_probs_contra — mccaskill_algo.rs:612-723 (without the final map) */
static int o_get_basepair_probs_contra(const rnamc_params* p, const uint8_t* seq, ostate* s,
                                       int allows_short_hairpins) {
  const rnamc_fold_score_sets* f = &p->contra;
  const uint32_t n = s->n;
  const Score global_sum = s->sums_external[IDX(0, n - 1)];
  const uint32_t min_len = allows_short_hairpins ? 2 : RNAMC_MIN_SPAN_HAIRPIN_CLOSE;
  for (uint32_t subseq_len = n; subseq_len >= min_len && subseq_len >= 1; subseq_len--) {
    for (uint32_t i = 0; i + subseq_len <= n; i++) {
      uint32_t j = i + subseq_len - 1;
      Score sum = ONEG_INF;
      Score sum2 = sum;
      for (uint32_t k = j + 1; k < n; k++) {
        Score x = s->sums_close[IDX(i, k)];
        if (x > ONEG_INF) {
          Score basepair_prob = s->basepair_probs[IDX(i, k)];
          if (!(basepair_prob > ONEG_INF)) return 100;
          Score multibranch_close_score = s->multibranch_close_scores[IDX(i, k)];
          x = basepair_prob + multibranch_close_score - x;
          o_logsumexp(&sum, x + s->sums_1ormore_basepairs[IDX(j + 1, k - 1)]);
          o_logsumexp(&sum2, x + f->multibranch_score_unpair * (Score)(k - j - 1));
        }
      }
      s->probs_multibranch[IDX(i, j)] = sum;
      s->probs_multibranch2[IDX(i, j)] = sum2;
      Score sum_close = s->sums_close[IDX(i, j)];
      if (sum_close > ONEG_INF) {
        Score sum_pair0 = (i < 1) ? 0.f : s->sums_external[IDX(0, i - 1)];
        Score sum_pair1 = (j > n - 2) ? 0.f : s->sums_external[IDX(j + 1, n - 1)];
        sum = sum_pair0 + sum_pair1 + s->sums_accessible[IDX(i, j)] + f->external_score_basepair -
              global_sum;
        for (uint32_t kk = i; kk-- > 0;) {
          uint32_t k = kk;
          if (i - k - 1 > RNAMC_MAX_LOOP_LEN) break;
          for (uint32_t l = j + 1; l < n; l++) {
            if (l - j - 1 + i - k - 1 > RNAMC_MAX_LOOP_LEN) break;
            Score x = s->sums_close[IDX(k, l)];
            if (x > ONEG_INF) {
              Score bp = s->basepair_probs[IDX(k, l)];
              if (!(bp > ONEG_INF)) return 100;
              o_logsumexp(&sum, bp + sum_close - x + o_get_2loop_score_contra(f, seq, k, l, i, j));
            }
          }
        }
        Score sum_accessible = s->sums_accessible[IDX(i, j)] + f->multibranch_score_basepair;
        for (uint32_t k = 0; k < i; k++) {
          Score x = s->sums_1ormore_basepairs[IDX(k + 1, i - 1)];
          o_logsumexp(&sum, sum_accessible + s->probs_multibranch2[IDX(k, j)] + x);
          Score y = s->probs_multibranch[IDX(k, j)];
          o_logsumexp(&sum, sum_accessible + y + f->multibranch_score_unpair * (Score)(i - k - 1));
          o_logsumexp(&sum, sum_accessible + x + y);
        }
        if (sum > ONEG_INF) s->basepair_probs[IDX(i, j)] = sum;
      }
    }
  }
  return 0;
}

static int check_args(const rnamc_params* p, const uint8_t* seq, uint32_t n) {
  if (!p || !seq) return RNAMC_ERR_INVALID_ARG;
  if (p->abi_version != RNAMC_ABI_VERSION || p->struct_bytes != sizeof(rnamc_params))
    return RNAMC_ERR_INVALID_ARG;
  if (n == 0) return RNAMC_ERR_EMPTY_SEQ; /* reference panics at mccaskill_algo.rs:526/622 */
  if (n > RNAMC_MAX_SEQ_LEN) return RNAMC_ERR_SEQ_TOO_LONG;
  for (uint32_t x = 0; x < n; x++)
    if (seq[x] > 3) return RNAMC_ERR_INVALID_BASE;
  return RNAMC_OK;
}

#ifndef ORACLE_EXACT /* the exported entry points of the reference restatement */
/* mccaskill_algo — mccaskill_algo.rs:247-280.  `mats`, if non-NULL, receives
 * copies of DP matrices (each n*n row-major, caller-allocated, entries may be
 * NULL) in the order of rnamc_debug_fetch's `which`. */
int rnamc_oracle_bpp_dump(const rnamc_params* p, const uint8_t* seq, uint32_t n,
                          int uses_contra_model, int allows_short_hairpins, float* bpp_packed,
                          float* log_partition, float** mats) {
  int st = check_args(p, seq, n);
  if (st) return st;
  ostate s;
  st = ostate_init(&s, n);
  if (st) return st;
  int bad;
  if (uses_contra_model) {
    o_get_fold_sums_contra(p, seq, &s, allows_short_hairpins);
    bad = o_get_basepair_probs_contra(p, seq, &s, allows_short_hairpins);
  } else {
    o_get_fold_sums(p, seq, &s);
    bad = o_get_basepair_probs(p, seq, &s);
  }
  if (bad) {
    ostate_free(&s);
    return bad;
  }
  if (log_partition) *log_partition = s.sums_external[IDX(0, n - 1)];
  if (bpp_packed) {
    /* final map, mccaskill_algo.rs:608/721; packed diagonal-major, absent = -1 */
    size_t o = 0;
    for (uint32_t d = 0; d < n; d++)
      for (uint32_t i = 0; i + d < n; i++) {
        Score lp = s.basepair_probs[IDX(i, i + d)];
        bpp_packed[o++] = (lp > ONEG_INF) ? o_expf(lp) : -1.0f;
      }
  }
  if (mats) {
    const Score* src[7] = {s.sums_close,
                           s.sums_accessible,
                           s.sums_external,
                           s.sums_1ormore_basepairs,
                           s.multibranch_close_scores,
                           s.probs_multibranch,
                           s.probs_multibranch2};
    for (int m = 0; m < 7; m++)
      if (mats[m])
        for (size_t x = 0; x < (size_t)n * n; x++) mats[m][x] = (float)src[m][x];
  }
  ostate_free(&s);
  return RNAMC_OK;
}

/* get_fold_sums / get_fold_sums_contra alone — mccaskill_algo.rs:282-378 / 380-516: the seven
 * members of FoldSums<T> (3-11) as the first stage returns them, each n*n row-major
 * (caller-allocated, entries may be NULL), in the struct's order: sums_external,
 * sums_rightmost_basepairs_external, sums_rightmost_basepairs_multibranch, sums_close,
 * sums_accessible, sums_multibranch, sums_1ormore_basepairs.  The sparse maps are dense here with
 * -inf for an absent key (value-preserving: SURVEY 8a N3). */
int rnamc_oracle_fold_sums(const rnamc_params* p, const uint8_t* seq, uint32_t n,
                           int uses_contra_model, int allows_short_hairpins, float** mats) {
  int st = check_args(p, seq, n);
  if (st) return st;
  if (!mats) return RNAMC_ERR_INVALID_ARG;
  ostate s;
  st = ostate_init(&s, n);
  if (st) return st;
  if (uses_contra_model)
    o_get_fold_sums_contra(p, seq, &s, allows_short_hairpins);
  else
    o_get_fold_sums(p, seq, &s);
  const Score* src[7] = {s.sums_external,
                         s.sums_rightmost_basepairs_external,
                         s.sums_rightmost_basepairs_multibranch,
                         s.sums_close,
                         s.sums_accessible,
                         s.sums_multibranch,
                         s.sums_1ormore_basepairs};
  for (int m = 0; m < 7; m++)
    if (mats[m])
      for (size_t x = 0; x < (size_t)n * n; x++) mats[m][x] = (float)src[m][x];
  ostate_free(&s);
  return RNAMC_OK;
}

int rnamc_oracle_bpp(const rnamc_params* p, const uint8_t* seq, uint32_t n, int uses_contra_model,
                     int allows_short_hairpins, float* bpp_packed, float* log_partition) {
  return rnamc_oracle_bpp_dump(p, seq, n, uses_contra_model, allows_short_hairpins, bpp_packed,
                               log_partition, NULL);
}

/* FoldScores<T> as the reference's inside pass leaves it (mccaskill_algo.rs:14-19; inserts
 * at 302-304, 320, 333-338 / 407-409, 431, 457-462).  Outputs in the layout of
 * rnamc_fold_scores (include/rnamc.h): three packed diagonal-major triangles with NaN =
 * key absent, and the twoloop_scores inserts in the reference's own visiting order.
 * *twoloop is malloc'ed here; release it with rnamc_oracle_free. */
int rnamc_oracle_fold_scores(const rnamc_params* p, const uint8_t* seq, uint32_t n,
                             int uses_contra_model, int allows_short_hairpins,
                             float* hairpin_scores, float* multibranch_close_scores,
                             float* accessible_scores, rnamc_twoloop_score** twoloop,
                             uint64_t* twoloop_count) {
  int st = check_args(p, seq, n);
  if (st) return st;
  ostate s;
  st = ostate_init(&s, n);
  if (st) return st;
  s.rec_hairpin = alloc_fill((size_t)n * n, NAN);
  s.rec_accessible = alloc_fill((size_t)n * n, NAN);
  if (!s.rec_hairpin || !s.rec_accessible) {
    ostate_free(&s);
    return RNAMC_ERR_OOM;
  }
  if (uses_contra_model)
    o_get_fold_sums_contra(p, seq, &s, allows_short_hairpins);
  else
    o_get_fold_sums(p, seq, &s);
  if (s.rec_failed) {
    ostate_free(&s);
    return RNAMC_ERR_OOM;
  }
  size_t o = 0;
  for (uint32_t d = 0; d < n; d++)
    for (uint32_t i = 0; i + d < n; i++, o++) {
      const size_t x = IDX(i, i + d);
      const int member = s.sums_close[x] > ONEG_INF;
      if (hairpin_scores) hairpin_scores[o] = s.rec_hairpin[x];
      if (multibranch_close_scores)
        multibranch_close_scores[o] = member ? s.multibranch_close_scores[x] : NAN;
      if (accessible_scores) accessible_scores[o] = member ? s.rec_accessible[x] : NAN;
    }
  if (twoloop_count) *twoloop_count = s.rec_count;
  if (twoloop) {
    *twoloop = s.rec_twoloop;
    s.rec_twoloop = NULL;
  }
  ostate_free(&s);
  return RNAMC_OK;
}

void rnamc_oracle_free(void* ptr) { free(ptr); }

/* ---- batch on a thread pool: one sequence per task, as the reference's
 * binaries do with scoped_threadpool (src/bin/mccaskill_algo.rs:58-93). ---- */
typedef struct {
  const rnamc_params* p;
  uint32_t n_seqs;
  const uint8_t* bases;
  const uint64_t* offsets;
  int contra, shorthp;
  float* bpp;
  const uint64_t* out_offsets;
  float* logz;
  volatile uint32_t next;
  volatile int status;
  pthread_mutex_t mu;
} obatch;

static void* obatch_worker(void* arg) {
  obatch* b = (obatch*)arg;
  for (;;) {
    pthread_mutex_lock(&b->mu);
    uint32_t s = b->next++;
    pthread_mutex_unlock(&b->mu);
    if (s >= b->n_seqs) break;
    uint32_t n = (uint32_t)(b->offsets[s + 1] - b->offsets[s]);
    int st = rnamc_oracle_bpp(b->p, b->bases + b->offsets[s], n, b->contra, b->shorthp,
                              b->bpp ? b->bpp + b->out_offsets[s] : NULL,
                              b->logz ? b->logz + s : NULL);
    if (st) {
      pthread_mutex_lock(&b->mu);
      if (!b->status) b->status = st;
      pthread_mutex_unlock(&b->mu);
    }
  }
  return NULL;
}

int rnamc_oracle_bpp_batch(const rnamc_params* p, uint32_t n_seqs, const uint8_t* bases,
                           const uint64_t* offsets, int uses_contra_model,
                           int allows_short_hairpins, float* bpp, const uint64_t* out_offsets,
                           float* log_partition, uint32_t n_threads) {
  if (!p || !bases || !offsets || (bpp && !out_offsets)) return RNAMC_ERR_INVALID_ARG;
  if (n_threads == 0) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  obatch b;
  memset(&b, 0, sizeof(b));
  b.p = p;
  b.n_seqs = n_seqs;
  b.bases = bases;
  b.offsets = offsets;
  b.contra = uses_contra_model;
  b.shorthp = allows_short_hairpins;
  b.bpp = bpp;
  b.out_offsets = out_offsets;
  b.logz = log_partition;
  pthread_mutex_init(&b.mu, NULL);
  pthread_t th[256];
  uint32_t started = 0;
  for (uint32_t x = 0; x + 1 < n_threads; x++) {
    if (pthread_create(&th[started], NULL, obatch_worker, &b) == 0) started++;
  }
  obatch_worker(&b);
  for (uint32_t x = 0; x < started; x++) pthread_join(th[x], NULL);
  pthread_mutex_destroy(&b.mu);
  return b.status;
}

/* centroid_fold — src/centroid_fold.rs:25-105, driven off a packed bpp triangle
 * (absent = negative).  pairs in push order. */
int rnamc_oracle_centroid_fold(const float* bpp_packed, uint32_t n, float centroid_threshold,
                               uint32_t* pairs_out, uint32_t max_pairs, uint32_t* n_pairs,
                               float* expect_accuracy) {
  if (!bpp_packed || n == 0 || !n_pairs) return RNAMC_ERR_INVALID_ARG;
  float* m = (float*)calloc((size_t)n * n, sizeof(float));
  if (!m) return RNAMC_ERR_OOM;
#define BPP(i, j) bpp_packed[(size_t)((j) - (i)) * n - (size_t)((j) - (i)) * ((j) - (i)-1) / 2 + (i)]
  for (uint32_t subseq_len = 1; subseq_len <= n; subseq_len++) {
    for (uint32_t i = 0; i + subseq_len <= n; i++) {
      uint32_t j = i + subseq_len - 1;
      if (i == j) continue;
      float best = m[IDX(i + 1, j)];
      float ea = m[IDX(i, j - 1)];
      if (ea > best) best = ea;
      float x = BPP(i, j);
      if (x >= -0.5f) { /* map hit */
        /* max[i+1][j-1]: for j == i+1 this is the lower-triangle zero */
        ea = m[IDX(i + 1, j - 1)] + centroid_threshold * x - 1.f;
        if (ea > best) best = ea;
      }
      for (uint32_t k = i + 1; k < j; k++) {
        ea = m[IDX(i, k)] + m[IDX(k + 1, j)];
        if (ea > best) best = ea;
      }
      m[IDX(i, j)] = best;
    }
  }
  uint32_t np = 0;
  /* traceback with an explicit stack, centroid_fold.rs:64-102 */
  uint32_t cap = 2 * n + 4, sp = 0;
  int64_t* stack = (int64_t*)malloc(sizeof(int64_t) * 2 * cap);
  if (!stack) {
    free(m);
    return RNAMC_ERR_OOM;
  }
  stack[0] = 0;
  stack[1] = (int64_t)n - 1;
  sp = 1;
  while (sp > 0) {
    sp--;
    int64_t i = stack[2 * sp], j = stack[2 * sp + 1];
    if (j <= i) continue;
    float best = m[IDX(i, j)];
    if (best == 0.f) continue;
    if (best == m[IDX(i + 1, j)]) {
      stack[2 * sp] = i + 1, stack[2 * sp + 1] = j, sp++;
    } else if (best == m[IDX(i, j - 1)]) {
      stack[2 * sp] = i, stack[2 * sp + 1] = j - 1, sp++;
    } else if (BPP(i, j) >= -0.5f &&
               best == m[IDX(i + 1, j - 1)] + centroid_threshold * BPP(i, j) - 1.f) {
      stack[2 * sp] = i + 1, stack[2 * sp + 1] = j - 1, sp++;
      if (pairs_out && np < max_pairs) {
        pairs_out[2 * np] = (uint32_t)i;
        pairs_out[2 * np + 1] = (uint32_t)j;
      }
      np++;
    } else {
      for (int64_t k = i + 1; k < j; k++) {
        if (best == m[IDX(i, k)] + m[IDX(k + 1, j)]) {
          stack[2 * sp] = i, stack[2 * sp + 1] = k, sp++;
          stack[2 * sp] = k + 1, stack[2 * sp + 1] = j, sp++;
          break;
        }
      }
    }
  }
#undef BPP
  *n_pairs = np;
  if (expect_accuracy) *expect_accuracy = m[IDX(0, n - 1)];
  free(stack);
  free(m);
  return RNAMC_OK;
}
#endif /* !ORACLE_EXACT */
