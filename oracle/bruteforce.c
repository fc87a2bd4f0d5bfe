/*
 * bruteforce.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Exhaustive f64 check of what the reference's recurrences MEAN: enumerate every
 * admissible secondary structure of a short sequence, score it by loop
 * decomposition through the same scoring functions (oracle_scoring.h), and sum
 * Boltzmann weights exactly.  Independent of the DP's index algebra, so it pins
 * the restatement's loop bounds and term structure.
 *
 * Structure space implied by /root/reference/src/mccaskill_algo.rs:
 *  - pairs are canonical; Turner: span >= 5 (282-300); CONTRAfold: span >= 5 or
 *    allows_short_hairpins (401-403);
 *  - a pair closes a hairpin (CONTRAfold: only if loop <= MAX_LOOP_LEN, 405-411),
 *    a 2-loop with <= 30 unpaired (306-325 / 412-436) or a multiloop with >= 2
 *    branches (326-330 / 437-448 via sums_multibranch, 364-374 / 499-512);
 *  - Turner multiloop: mbclose + sum over branches (accessible + COEFF_NUM_BRANCHES),
 *    unpaired free; exterior: sum over branches accessible, unpaired free;
 *  - CONTRAfold multiloop: mb_base + mb_bp + junction(i,j) + per branch
 *    (accessible + mb_bp) + mb_unpair per unpaired base; exterior: per branch
 *    (accessible + ext_bp) + ext_unpair per unpaired base (468-512).
 */
#include <stdlib.h>

#include "oracle_scoring.h"

typedef struct {
  const rnamc_params* p;
  const uint8_t* seq;
  uint32_t n;
  int contra, shorthp;
  int pt[64];
  double z;
  double* bp; /* n*n */
  uint64_t count;
} bf;

static int bf_allowed(const bf* b, int i, int j) {
  if (!o_has_canonical_basepair(b->seq[i], b->seq[j])) return 0;
  if (b->contra && b->shorthp) return 1;
  return j - i + 1 >= RNAMC_MIN_SPAN_HAIRPIN_CLOSE;
}

/* score of the loop closed by (i,j); returns 0 and sets *ok=0 when inadmissible */
static double bf_loop(const bf* b, int i, int j, int* ok) {
  int br[64][2], nb = 0, unp = 0;
  for (int q = i + 1; q < j;) {
    if (b->pt[q] > q) {
      br[nb][0] = q, br[nb][1] = b->pt[q], nb++;
      q = b->pt[q] + 1;
    } else {
      unp++, q++;
    }
  }
  const uint8_t* s = b->seq;
  uint32_t n = b->n;
  double sc = 0;
  if (nb == 0) {
    if (b->contra) {
      if (j - i - 1 > RNAMC_MAX_LOOP_LEN) {
        *ok = 0;
        return 0;
      }
      sc = o_get_hairpin_score_contra(&b->p->contra, s, i, j);
    } else {
      sc = o_get_hairpin_score(&b->p->turner, s, i, j);
    }
  } else if (nb == 1) {
    if (unp > RNAMC_MAX_2LOOP_LEN) {
      *ok = 0;
      return 0;
    }
    sc = b->contra ? o_get_2loop_score_contra(&b->p->contra, s, i, j, br[0][0], br[0][1])
                   : o_get_2loop_score(&b->p->turner, s, i, j, br[0][0], br[0][1]);
  } else {
    if (b->contra) {
      const rnamc_fold_score_sets* f = &b->p->contra;
      sc = o_get_multibranch_close_score_contra(f, s, n, i, j);
      sc += (double)f->multibranch_score_unpair * unp;
      for (int x = 0; x < nb; x++)
        sc += (double)o_get_accessible_score_contra(f, s, n, br[x][0], br[x][1]) +
              (double)f->multibranch_score_basepair;
    } else {
      const rnamc_turner_scores* t = &b->p->turner;
      sc = o_get_multibranch_close_score(t, s, i, j);
      for (int x = 0; x < nb; x++)
        sc += (double)o_get_accessible_score(t, s, n, br[x][0], br[x][1]) +
              (double)t->coeff_num_branches;
    }
  }
  for (int x = 0; x < nb && *ok; x++) sc += bf_loop(b, br[x][0], br[x][1], ok);
  return sc;
}

static void bf_eval(bf* b) {
  int ok = 1;
  double sc = 0;
  const uint8_t* s = b->seq;
  uint32_t n = b->n;
  for (int q = 0; q < (int)n;) {
    if (b->pt[q] > q) {
      int l = b->pt[q];
      if (b->contra) {
        sc += (double)o_get_accessible_score_contra(&b->p->contra, s, n, q, l) +
              (double)b->p->contra.external_score_basepair;
      } else {
        sc += (double)o_get_accessible_score(&b->p->turner, s, n, q, l);
      }
      sc += bf_loop(b, q, l, &ok);
      if (!ok) return;
      q = l + 1;
    } else {
      if (b->contra) sc += (double)b->p->contra.external_score_unpair;
      q++;
    }
  }
  double w = exp(sc);
  b->z += w;
  b->count++;
  for (int q = 0; q < (int)n; q++)
    if (b->pt[q] > q) b->bp[(size_t)q * n + b->pt[q]] += w;
}

static void bf_rec(bf* b, int pos) {
  int n = (int)b->n;
  if (pos == n) {
    bf_eval(b);
    return;
  }
  if (b->pt[pos] != -1) { /* closing partner of an earlier base */
    bf_rec(b, pos + 1);
    return;
  }
  bf_rec(b, pos + 1); /* unpaired */
  int bound = n;      /* innermost enclosing pair's closing position */
  for (int a = 0; a < pos; a++)
    if (b->pt[a] > pos && b->pt[a] < bound) bound = b->pt[a];
  for (int q = pos + 1; q < bound; q++) {
    if (b->pt[q] != -1 || !bf_allowed(b, pos, q)) continue;
    b->pt[pos] = q;
    b->pt[q] = pos;
    bf_rec(b, pos + 1);
    b->pt[pos] = -1;
    b->pt[q] = -1;
  }
}

/* log_z: ln of the exact partition function; bpp_full: n*n exact pair
 * probabilities (upper triangle); n_structs: number of admissible structures. */
int rnamc_oracle_bruteforce(const rnamc_params* p, const uint8_t* seq, uint32_t n,
                            int uses_contra_model, int allows_short_hairpins, double* log_z,
                            double* bpp_full, uint64_t* n_structs) {
  if (!p || !seq || n == 0 || n > 40) return RNAMC_ERR_INVALID_ARG;
  bf b;
  memset(&b, 0, sizeof(b));
  b.p = p;
  b.seq = seq;
  b.n = n;
  b.contra = uses_contra_model;
  b.shorthp = allows_short_hairpins;
  for (int x = 0; x < 64; x++) b.pt[x] = -1;
  b.bp = (double*)calloc((size_t)n * n, sizeof(double));
  if (!b.bp) return RNAMC_ERR_OOM;
  bf_rec(&b, 0);
  if (log_z) *log_z = log(b.z);
  if (bpp_full)
    for (size_t x = 0; x < (size_t)n * n; x++) bpp_full[x] = b.bp[x] / b.z;
  if (n_structs) *n_structs = b.count;
  free(b.bp);
  return RNAMC_OK;
}
