// rnamc_scoring.h — loop-score functions of both models, usable from device code (the
// sweep kernels) and from host code (FoldScores materialisation).  Same expression trees
// as the reference (src/utils.rs:162-556, src/mccaskill_algo.rs:437-455); compile with
// -ffp-contract=off.
#ifndef RNAMC_SCORING_H
#define RNAMC_SCORING_H

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/rnamc.h"

#define RNAMC_HD __host__ __device__ __forceinline__

namespace rnamc {

constexpr float kNegInf = -__builtin_inff();

RNAMC_HD bool canonical(int a, int b) {
  // AU CG GC GU UA UG  <=>  a+b == 3 (AU, CG) or {a,b} == {G,U}
  return (a + b == 3) || (a + b == 5);
}

RNAMC_HD bool augu(int a, int b) {
  // AU UA GU UG: canonical and not CG/GC
  return canonical(a, b) && !((a == 1 && b == 2) || (a == 2 && b == 1));
}

// ----------------------------------------------------------------------------
// Turner model scores: src/utils.rs:166-411

struct Turner {
  const rnamc_turner_scores& t;
  const float* hp_init;  // hairpin initiation by loop length incl. extrapolation

  RNAMC_HD float pen(int a, int b) const {
    return augu(a, b) ? t.helix_augu_end_penalty : 0.f;
  }

  // get_hairpin_score, src/utils.rs:166-205
  RNAMC_HD float hairpin(const uint8_t* s, uint32_t /*n*/, uint32_t i, uint32_t j) const {
    const uint32_t span = j - i + 1;
    if (span <= RNAMC_MAX_SPECIAL_HAIRPIN_LEN) {
      for (uint32_t x = 0; x < t.num_special_hairpins; x++) {
        if (t.special_hairpin_lens[x] != span) continue;
        bool eq = true;
        for (uint32_t y = 0; y < span; y++) eq = eq && (t.special_hairpin_seqs[x][y] == s[i + y]);
        if (eq) {
          const float sc = t.special_hairpin_scores[x];
          if (sc > kNegInf) return sc;
          break;
        }
      }
    }
    const uint32_t len = j - i - 1;
    const int bi = s[i], bj = s[j];
    float hs;
    if (len == t.min_hairpin_len) {
      hs = hp_init[len];
    } else {
      hs = hp_init[len] + t.terminal_mismatch_scores_hairpin[bi][bj][s[i + 1]][s[j - 1]];
    }
    return hs + pen(bi, bj);
  }

  // get_2loop_score, src/utils.rs:207-366.  (i,j) closes, (k,l) is enclosed;
  // a = k-i-1 and b = j-l-1 unpaired bases on the two sides.  The sweep kernels use the
  // table-driven form of rnamc_probes.h (same trees); this one serves FoldScores.
  RNAMC_HD float twoloop(const uint8_t* s, uint32_t i, uint32_t j, uint32_t k, uint32_t l) const {
    const uint32_t a = k - i - 1, b = j - l - 1;
    const int ci = s[i], cj = s[j], ak = s[k], al = s[l];
    if (a == 0 && b == 0) return t.stack_scores[ci][cj][ak][al];
    if (a == 0 || b == 0) {
      const uint32_t len = a + b;
      if (len == 1) return t.bulge_scores_init[1] + t.stack_scores[ci][cj][ak][al];
      return t.bulge_scores_init[len] + pen(ci, cj) + pen(ak, al);
    }
    if (a == 1 && b == 1) return t.interior_scores_1x1[ci][cj][s[i + 1]][s[j - 1]][ak][al];
    if (a == 1 && b == 2)
      return t.interior_scores_1x2[ci][cj][s[i + 1]][s[j - 1]][s[j - 2]][ak][al];
    if (a == 2 && b == 1)
      return t.interior_scores_1x2[al][ak][s[j - 1]][s[i + 2]][s[i + 1]][cj][ci];
    if (a == 2 && b == 2)
      return t.interior_scores_2x2[ci][cj][s[i + 1]][s[j - 1]][s[i + 2]][s[j - 2]][ak][al];
    const uint32_t diff = a > b ? a - b : b - a;
    const int m0 = s[i + 1], m1 = s[j - 1], m2 = s[l + 1], m3 = s[k - 1];
    float mm;
    if (a == 1 || b == 1) {
      mm = t.terminal_mismatch_scores_1xmany[ci][cj][m0][m1] +
           t.terminal_mismatch_scores_1xmany[al][ak][m2][m3];
    } else if ((a == 2 && b == 3) || (a == 3 && b == 2)) {
      mm = t.terminal_mismatch_scores_2x3[ci][cj][m0][m1] +
           t.terminal_mismatch_scores_2x3[al][ak][m2][m3];
    } else {
      mm = t.terminal_mismatch_scores_interior[ci][cj][m0][m1] +
           t.terminal_mismatch_scores_interior[al][ak][m2][m3];
    }
    const float nin = t.ninio_coeff * static_cast<float>(diff);
    return t.interior_scores_init[a + b] + (nin > t.ninio_max ? nin : t.ninio_max) + mm +
           pen(ci, cj) + pen(ak, al);
  }

  // get_multibranch_close_score, src/utils.rs:368-382
  RNAMC_HD float mbclose(const uint8_t* s, uint32_t /*n*/, uint32_t i, uint32_t j) const {
    const int ci = s[i], cj = s[j];
    return t.init_multibranch_base +
           t.terminal_mismatch_scores_multibranch[cj][ci][s[j - 1]][s[i + 1]] + pen(ci, cj);
  }

  // get_accessible_score, src/utils.rs:384-411, uses_sentinel_bases = false
  RNAMC_HD float accessible(const uint8_t* s, uint32_t n, uint32_t i, uint32_t j) const {
    const int ai = s[i], aj = s[j];
    float sc;
    if (i > 0 && j < n - 1) {
      sc = t.terminal_mismatch_scores_multibranch[ai][aj][s[i - 1]][s[j + 1]];
    } else if (i > 0) {
      sc = t.dangling_scores_5prime[ai][aj][s[i - 1]];
    } else if (j < n - 1) {
      sc = t.dangling_scores_3prime[ai][aj][s[j + 1]];
    } else {
      sc = 0.f;
    }
    return sc + pen(ai, aj);
  }
};

// ----------------------------------------------------------------------------
// CONTRAfold model scores: src/utils.rs:413-556, src/mccaskill_algo.rs:437-455

struct Contra {
  const rnamc_fold_score_sets& f;

  RNAMC_HD float junction_single(const uint8_t* s, uint32_t y0, uint32_t y1) const {
    const int a0 = s[y0], a1 = s[y1];
    return f.helix_close_scores[a0][a1] + f.terminal_mismatch_scores[a0][a1][s[y0 + 1]][s[y1 - 1]];
  }

  RNAMC_HD float junction(const uint8_t* s, uint32_t n, uint32_t p0, uint32_t p1) const {
    const int b0 = s[p0], b1 = s[p1];
    return f.helix_close_scores[b0][b1] +
           (p0 < n - 1 ? f.dangling_scores_left[b0][b1][s[p0 + 1]] : 0.f) +
           (p1 > 0 ? f.dangling_scores_right[b0][b1][s[p1 - 1]] : 0.f);
  }

  RNAMC_HD float hairpin(const uint8_t* s, uint32_t /*n*/, uint32_t i, uint32_t j) const {
    uint32_t len = j - i - 1;
    if (len > RNAMC_MAX_LOOP_LEN) len = RNAMC_MAX_LOOP_LEN;
    return f.hairpin_scores_len_cumulative[len] + junction_single(s, i, j);
  }

  // get_2loop_score_contra, src/utils.rs:423-520 (+ base-pair score, mccaskill_algo.rs:437-455)
  RNAMC_HD float twoloop(const uint8_t* s, uint32_t i, uint32_t j, uint32_t k, uint32_t l) const {
    const uint32_t a = k - i - 1, b = j - l - 1;
    const int ak = s[k], al = s[l];
    float sc;
    if (a == 0 && b == 0) {
      sc = f.stack_scores[s[i]][s[j]][ak][al];
    } else if (a == 0 || b == 0) {
      const uint32_t len = a + b;
      float s0 = 0.f;
      if (len == 1) s0 = f.bulge_scores_0x1[a == 1 ? s[i + 1] : s[j - 1]];
      sc = s0 + f.bulge_scores_len_cumulative[len - 1] + junction_single(s, i, j) +
           junction_single(s, l, k);
    } else {
      float s0;
      if (a == b) {
        const float s11 = (a + b == 2) ? f.interior_scores_1x1[s[i + 1]][s[j - 1]] : 0.f;
        s0 = s11 + f.interior_scores_symmetric_cumulative[a - 1];
      } else {
        const uint32_t diff = a > b ? a - b : b - a;
        s0 = f.interior_scores_asymmetric_cumulative[diff - 1];
      }
      const float se = (a <= RNAMC_MAX_INTERIOR_EXPLICIT && b <= RNAMC_MAX_INTERIOR_EXPLICIT)
                           ? f.interior_scores_explicit[a - 1][b - 1]
                           : 0.f;
      sc = s0 + se + f.interior_scores_len_cumulative[a + b - 2] + junction_single(s, i, j) +
           junction_single(s, l, k);
    }
    return sc + f.basepair_scores[ak][al];
  }

  RNAMC_HD float mbclose(const uint8_t* s, uint32_t n, uint32_t i, uint32_t j) const {
    return f.multibranch_score_base + f.multibranch_score_basepair + junction(s, n, i, j);
  }

  RNAMC_HD float accessible(const uint8_t* s, uint32_t n, uint32_t i, uint32_t j) const {
    return junction(s, n, j, i) + f.basepair_scores[s[i]][s[j]];
  }
};


// ----------------------------------------------------------------------------
// Flat forms of the 2-loop scores: every table lookup of a call is issued at once (the branchy
// forms above walk the loop classes one after the other, and lanes of one wave hold different
// classes).  Arguments are base codes: (ci,cj) closes, (ak,al) is enclosed; x1,x2 = the two
// bases inside ci, y1,y2 inside cj; m2,m3 = the bases outside al and ak; a, b = unpaired bases
// on the two sides.  turner_twoloop_flat evaluates the same expression tree per class as
// Turner::twoloop (bit-identical; the reference-order latency forms rely on that).
RNAMC_HD float turner_twoloop_flat(const rnamc_turner_scores& t, uint32_t a,
                                                     uint32_t b, int ci, int cj, int x1, int x2,
                                                     int y1, int y2, int ak, int al, int m2, int m3) {
  const bool bulge = (a == 0u) != (b == 0u);
  const uint32_t len = a + b;
  const bool stack = len == 0u, bulge1 = bulge && len == 1u, bulgeN = bulge && len > 1u;
  const bool i11 = a == 1u && b == 1u, i12 = a == 1u && b == 2u, i21 = a == 2u && b == 1u,
             i22 = a == 2u && b == 2u;
  const bool generic = !(stack || bulge || i11 || i12 || i21 || i22);
  // primary entry
  const float* tmx = (a == 1u || b == 1u) ? &t.terminal_mismatch_scores_1xmany[0][0][0][0]
                     : ((a == 2u && b == 3u) || (a == 3u && b == 2u))
                         ? &t.terminal_mismatch_scores_2x3[0][0][0][0]
                         : &t.terminal_mismatch_scores_interior[0][0][0][0];
  const float* pa;
  if (stack || bulge1) {
    pa = &t.stack_scores[ci][cj][ak][al];
  } else if (i11) {
    pa = &t.interior_scores_1x1[ci][cj][x1][y1][ak][al];
  } else if (i12) {
    pa = &t.interior_scores_1x2[ci][cj][x1][y1][y2][ak][al];
  } else if (i21) {
    pa = &t.interior_scores_1x2[al][ak][y1][x2][x1][cj][ci];
  } else if (i22) {
    pa = &t.interior_scores_2x2[ci][cj][x1][y1][x2][y2][ak][al];
  } else {
    pa = tmx + ((ci * 4 + cj) * 4 + x1) * 4 + y1;  // (bulgeN: read, not used)
  }
  const float* pb = tmx + ((al * 4 + ak) * 4 + m2) * 4 + m3;
  const float* pc = bulge ? &t.bulge_scores_init[len] : &t.interior_scores_init[len];
  const float A = *pa, B = *pb, C = *pc;  // three independent loads
  const float penc = augu(ci, cj) ? t.helix_augu_end_penalty : 0.f;
  const float peni = augu(ak, al) ? t.helix_augu_end_penalty : 0.f;
  if (generic) {
    const uint32_t diff = a > b ? a - b : b - a;
    const float nin = t.ninio_coeff * static_cast<float>(diff);
    const float mm = A + B;
    return C + (nin > t.ninio_max ? nin : t.ninio_max) + mm + penc + peni;
  }
  if (bulge1) return C + A;
  if (bulgeN) return C + penc + peni;
  return A;
}

// get_2loop_score_contra (src/utils.rs:423-520) + the enclosed pair's base-pair score
// (src/mccaskill_algo.rs:441).  Same terms as Contra::twoloop; the association of the sum
// differs (tree-order mode only: not bit-comparable with the reference anyway).
RNAMC_HD float contra_twoloop_flat(const rnamc_fold_score_sets& f, uint32_t a, uint32_t b, int ci,
                                   int cj, int x1, int y1, int ak, int al, int m2, int m3) {
  const uint32_t len = a + b;
  const bool stack = len == 0u;
  const bool bulge = (a == 0u) != (b == 0u);
  const bool inter = a != 0u && b != 0u;
  const uint32_t diff = a > b ? a - b : b - a;
  // all independent loads, indices clamped into their tables
  const float st = f.stack_scores[ci][cj][ak][al];
  const float bp = f.basepair_scores[ak][al];
  const float js0 = f.helix_close_scores[ci][cj] + f.terminal_mismatch_scores[ci][cj][x1][y1];
  const float js1 = f.helix_close_scores[al][ak] + f.terminal_mismatch_scores[al][ak][m2][m3];
  const float b01 = f.bulge_scores_0x1[a == 1u ? x1 : y1];
  const uint32_t lb = len >= 1u ? (len - 1u < RNAMC_MAX_LOOP_LEN ? len - 1u : RNAMC_MAX_LOOP_LEN - 1u) : 0u;
  const float blen = f.bulge_scores_len_cumulative[lb];
  const uint32_t li = len >= 2u ? (len - 2u < RNAMC_MAX_LOOP_LEN - 1u ? len - 2u : RNAMC_MAX_LOOP_LEN - 2u) : 0u;
  const float ilen = f.interior_scores_len_cumulative[li];
  const float i11 = f.interior_scores_1x1[x1][y1];
  const uint32_t sa = a >= 1u ? (a - 1u < RNAMC_MAX_INTERIOR_SYMMETRIC ? a - 1u : RNAMC_MAX_INTERIOR_SYMMETRIC - 1u) : 0u;
  const float sym = f.interior_scores_symmetric_cumulative[sa];
  const uint32_t da = diff >= 1u ? (diff - 1u < RNAMC_MAX_INTERIOR_ASYMMETRIC ? diff - 1u : RNAMC_MAX_INTERIOR_ASYMMETRIC - 1u) : 0u;
  const float asym = f.interior_scores_asymmetric_cumulative[da];
  const bool expl = inter && a <= RNAMC_MAX_INTERIOR_EXPLICIT && b <= RNAMC_MAX_INTERIOR_EXPLICIT;
  const float ex = f.interior_scores_explicit[expl ? a - 1u : 0u][expl ? b - 1u : 0u];
  float sc;
  if (stack) {
    sc = st;
  } else if (bulge) {
    sc = (len == 1u ? b01 : 0.f) + blen + js0 + js1;
  } else {
    const float s0 = (a == b) ? ((len == 2u ? i11 : 0.f) + sym) : asym;
    sc = s0 + (expl ? ex : 0.f) + ilen + js0 + js1;
  }
  return sc + bp;
}

}  // namespace rnamc

#endif
