/*
 * mccaskill_exact.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * NOT the reference's arithmetic.  The loops of mccaskill_oracle.c (the restatement of
 * /root/reference/src/mccaskill_algo.rs:282-723) compiled a second time with
 * Score = double, logsumexp = max + log1p(exp(min - max)) and expf = exp
 * (oracle_scoring.h, ORACLE_EXACT): the mathematically exact value of the recurrences the
 * reference evaluates approximately.  It is the checker of the HIP path's tree-order
 * summation mode (rnamc_ctx_set "summation_mode" 1), whose sums are order-free and so
 * cannot be compared bit for bit with the reference's left fold; it also measures how far
 * the reference-order result itself sits from the exact one.
 */
#define ORACLE_EXACT 1
#include "mccaskill_oracle.c"

/* bpp_packed: n(n+1)/2 doubles, diagonal-major, absent = -1; *log_partition: ln Z */
int rnamc_oracle_exact_bpp(const rnamc_params* p, const uint8_t* seq, uint32_t n,
                           int uses_contra_model, int allows_short_hairpins, double* bpp_packed,
                           double* log_partition) {
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
    size_t o = 0;
    for (uint32_t d = 0; d < n; d++)
      for (uint32_t i = 0; i + d < n; i++) {
        Score lp = s.basepair_probs[IDX(i, i + d)];
        bpp_packed[o++] = (lp > ONEG_INF) ? exp(lp) : -1.0;
      }
  }
  ostate_free(&s);
  return RNAMC_OK;
}
