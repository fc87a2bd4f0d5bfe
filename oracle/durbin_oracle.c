/*
 * durbin_oracle.c — CPU restatement of the reference's Durbin pair-HMM match probabilities.
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, never by the product (rna_algos_amd/).
 *
 * Follows /root/reference/src/durbin_algo.rs loop for loop:
 *   get_align_sums   79-199  (forward 82-139, backward 140-197)
 *   get_match_probs  201-242
 * with logsumexp / expf of src/utils.rs:579-655 (oracle_scoring.h).  f32 throughout, no FMA
 * contraction.  Unlike the McCaskill path, every constant of this path is in the reference
 * tree itself (src/compiled_align_scores.rs); the caller passes them in rnamc_align_scores.
 * PARITY UNPINNED all the same: the reference's only test of it is the range assertion
 * tests/tests.rs:45-80, repeated in tests/test_durbin_cpu.py.
 *
 * Sequences carry PSEUDO_BASE (= 4, src/utils.rs:122) at both ends, as every caller of the
 * reference builds them (tests/tests.rs:53-55, src/bin/durbin_algo.rs:49-51).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../include/rnamc.h"
#include "oracle_scoring.h"

#define AT(m, i, j) (m)[(size_t)(i) * n2 + (j)]

int rnamc_oracle_durbin(const rnamc_align_scores* sc, const uint8_t* a, uint32_t n1,
                        const uint8_t* b, uint32_t n2, float* match_probs /* n1*n2 */) {
  if (!sc || !a || !b || !match_probs || n1 < 2 || n2 < 2) return RNAMC_ERR_INVALID_ARG;
  const size_t cells = (size_t)n1 * n2;
  float* buf = (float*)malloc(sizeof(float) * cells * 6);
  if (!buf) return RNAMC_ERR_OOM;
  for (size_t x = 0; x < cells * 6; x++) buf[x] = ONEG_INF; /* AlignSums::new, 60-71 */
  float *fm = buf, *fi = buf + cells, *fd = buf + 2 * cells;
  float *bm = buf + 3 * cells, *bi = buf + 4 * cells, *bd = buf + 5 * cells;
  /* forward, 92-151 */
  for (uint32_t i = 0; i + 1 < n1; i++) {
    for (uint32_t j = 0; j + 1 < n2; j++) {
      if (i == 0 && j == 0) {
        AT(fm, i, j) = 0.f;
        continue;
      }
      if (i > 0 && j > 0) {
        Score sum = ONEG_INF;
        Score match_score = sc->match_scores[a[i]][b[j]];
        int begins_sum = (i - 1 == 0) && (j - 1 == 0);
        Score term = AT(fm, i - 1, j - 1) + (begins_sum ? sc->init_match_score : sc->match2match_score);
        o_logsumexp(&sum, term);
        term = AT(fi, i - 1, j - 1) + sc->match2insert_score;
        o_logsumexp(&sum, term);
        term = AT(fd, i - 1, j - 1) + sc->match2insert_score;
        o_logsumexp(&sum, term);
        AT(fm, i, j) = sum + match_score;
      }
      if (i > 0) {
        Score insert_score = sc->insert_scores[a[i]];
        int begins_sum = (i - 1 == 0) && (j == 0);
        Score sum = ONEG_INF;
        Score term = AT(fm, i - 1, j) + (begins_sum ? sc->init_insert_score : sc->match2insert_score);
        o_logsumexp(&sum, term);
        term = AT(fi, i - 1, j) + sc->insert_extend_score;
        o_logsumexp(&sum, term);
        AT(fi, i, j) = sum + insert_score;
      }
      if (j > 0) {
        Score insert_score = sc->insert_scores[b[j]];
        int begins_sum = (i == 0) && (j - 1 == 0);
        Score sum = ONEG_INF;
        Score term = AT(fm, i, j - 1) + (begins_sum ? sc->init_insert_score : sc->match2insert_score);
        o_logsumexp(&sum, term);
        term = AT(fd, i, j - 1) + sc->insert_extend_score;
        o_logsumexp(&sum, term);
        AT(fd, i, j) = sum + insert_score;
      }
    }
  }
  /* backward, 152-213 */
  for (uint32_t i = n1 - 1; i >= 1; i--) {
    for (uint32_t j = n2 - 1; j >= 1; j--) {
      if (i == n1 - 1 && j == n2 - 1) {
        AT(bm, i, j) = 0.f;
        continue;
      }
      if (i < n1 - 1 && j < n2 - 1) {
        Score sum = ONEG_INF;
        Score match_score = sc->match_scores[a[i]][b[j]];
        int ends_sum = (i + 1 == n1 - 1) && (j + 1 == n2 - 1);
        Score term = AT(bm, i + 1, j + 1) + (ends_sum ? 0.f : sc->match2match_score);
        o_logsumexp(&sum, term);
        term = AT(bi, i + 1, j + 1) + sc->match2insert_score;
        o_logsumexp(&sum, term);
        term = AT(bd, i + 1, j + 1) + sc->match2insert_score;
        o_logsumexp(&sum, term);
        AT(bm, i, j) = sum + match_score;
      }
      if (i < n1 - 1) {
        Score insert_score = sc->insert_scores[a[i]];
        int ends_sum = (i + 1 == n1 - 1) && (j == n2 - 1);
        Score sum = ONEG_INF;
        Score term = AT(bm, i + 1, j) + (ends_sum ? 0.f : sc->match2insert_score);
        o_logsumexp(&sum, term);
        term = AT(bi, i + 1, j) + sc->insert_extend_score;
        o_logsumexp(&sum, term);
        AT(bi, i, j) = sum + insert_score;
      }
      if (j < n2 - 1) {
        Score insert_score = sc->insert_scores[b[j]];
        int ends_sum = (i == n1 - 1) && (j + 1 == n2 - 1);
        Score sum = ONEG_INF;
        Score term = AT(bm, i, j + 1) + (ends_sum ? 0.f : sc->match2insert_score);
        o_logsumexp(&sum, term);
        term = AT(bd, i, j + 1) + sc->insert_extend_score;
        o_logsumexp(&sum, term);
        AT(bd, i, j) = sum + insert_score;
      }
    }
  }
  /* get_match_probs, 217-264 */
  for (size_t x = 0; x < cells; x++) match_probs[x] = 0.f;
  Score global_sum = AT(fm, n1 - 2, n2 - 2);
  o_logsumexp(&global_sum, AT(fi, n1 - 2, n2 - 2));
  o_logsumexp(&global_sum, AT(fd, n1 - 2, n2 - 2));
  for (uint32_t i = 1; i + 1 < n1; i++) {
    for (uint32_t j = 1; j + 1 < n2; j++) {
      Score sum = ONEG_INF;
      Score forward_sum = AT(fm, i, j);
      int ends_sum = (i + 1 == n1 - 1) && (j + 1 == n2 - 1);
      Score term = (ends_sum ? 0.f : sc->match2match_score) + AT(bm, i + 1, j + 1);
      o_logsumexp(&sum, term);
      term = sc->match2insert_score + AT(bi, i + 1, j + 1);
      o_logsumexp(&sum, term);
      term = sc->match2insert_score + AT(bd, i + 1, j + 1);
      o_logsumexp(&sum, term);
      AT(match_probs, i, j) = o_expf(forward_sum + sum - global_sum);
    }
  }
  free(buf);
  return RNAMC_OK;
}
