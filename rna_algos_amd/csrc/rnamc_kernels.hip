// rnamc_kernels.hip — hand-written gfx950 kernels of the McCaskill inside/outside
// sweep, reference-order ("bit-faithful") summation.
//
// What is computed (reference: src/mccaskill_algo.rs:282-723 of heartsh/rna-algos):
// the log-space partition-function recurrences of the Turner and CONTRAfold
// models and the outside recursion giving base-pairing probabilities.  Every
// `⊕` below is the reference's approximate, NON-associative `logsumexp`
// (src/utils.rs:579-627); to match the reference CPU path each per-cell fold is
// evaluated strictly in the reference's k order with separate f32 mul/add
// (this file must be compiled with -ffp-contract=off and without fast-math).
//
// Mapping (MI355X-first, not a translation of the Rust loops):
//  * the sweep is by anti-diagonal d = j - i; all cells of one diagonal of ALL
//    sequences of a group are independent and are processed by one launch;
//  * one lane owns one cell (i, i+d); lanes of a wave own consecutive i, so with
//    the diagonal-major / row-major packed layouts of rnamc_internal.h every
//    operand of every inner loop is a coalesced 256-B wave access;
//  * the reduction index k is walked sequentially per lane (order is part of
//    the result); parallelism comes from cells x sequences, not from k.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "rnamc_device.h"

namespace rnamc {

namespace {

constexpr float kNegInf = -__builtin_inff();

// ----------------------------------------------------------------------------
// numerics: src/utils.rs:579-655

__device__ __forceinline__ float ln_exp_1p(float x) {
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

// One fold step sum ⊕ x.  Operands are finite or -inf (never NaN/+inf: absent
// map entries are -inf and are masked at the source), so the reference's two
// is_finite() early-outs collapse to "if the smaller one is -inf take the
// larger one".
__device__ __forceinline__ float lse(float sum, float x) {
  float hi = fmaxf(sum, x);
  float lo = fminf(sum, x);
  float z = hi - lo;
  float r = lo + (z >= 11.862479f ? z : ln_exp_1p(z));
  return (lo == kNegInf) ? hi : r;
}

__device__ __forceinline__ float expf_ref(float x) {
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
    // libm exp for x >= 0 (src/utils.rs:653): evaluate in f64 and round once.
    return static_cast<float>(exp(static_cast<double>(x)));
  }
}

// ----------------------------------------------------------------------------
// index algebra

__device__ __forceinline__ uint32_t tri_off(uint32_t n, uint32_t d) {
  // start of diagonal d (diag-major) == start of row d (row-major)
  return d * n - (d * (d - 1u)) / 2u;
}

__device__ __forceinline__ bool canonical(int a, int b) {
  // AU CG GC GU UA UG  <=>  a+b == 3 (AU, CG) or {a,b} == {G,U}
  return (a + b == 3) || (a + b == 5);
}

__device__ __forceinline__ bool augu(int a, int b) {
  // AU UA GU UG: canonical and not CG/GC
  return canonical(a, b) && !((a == 1 && b == 2) || (a == 2 && b == 1));
}

struct Seq {
  const uint8_t* s;  // base codes
  uint32_t n;
  float* m[M_COUNT];
  float* out;  // packed bpp triangle (log domain until finalize)
};

__device__ __forceinline__ Seq load_seq(const DeviceBatch& b, uint32_t which) {
  const SeqDesc sd = b.seqs[which];
  Seq q;
  q.s = b.bases + sd.seq_off;
  q.n = sd.n;
  float* base = b.workspace + sd.ws_off;
#pragma unroll
  for (int x = 0; x < M_COUNT; x++) q.m[x] = base + static_cast<size_t>(x) * sd.tri_pad;
  q.out = b.out + sd.out_off;
  return q;
}

// ----------------------------------------------------------------------------
// Turner model scores: src/utils.rs:166-411

struct Turner {
  const rnamc_turner_scores& t;
  const float* hp_init;  // hairpin initiation by loop length incl. extrapolation

  __device__ __forceinline__ float pen(int a, int b) const {
    return augu(a, b) ? t.helix_augu_end_penalty : 0.f;
  }

  // get_hairpin_score, src/utils.rs:166-205
  __device__ float hairpin(const uint8_t* s, uint32_t /*n*/, uint32_t i, uint32_t j) const {
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
  // a = k-i-1 and b = j-l-1 unpaired bases on the two sides.
  __device__ float twoloop(const uint8_t* s, uint32_t i, uint32_t j, uint32_t k, uint32_t l,
                           uint32_t a, uint32_t b) const {
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
    return t.interior_scores_init[a + b] +
           fmaxf(t.ninio_coeff * static_cast<float>(diff), t.ninio_max) + mm + pen(ci, cj) +
           pen(ak, al);
  }

  // get_multibranch_close_score, src/utils.rs:368-382
  __device__ float mbclose(const uint8_t* s, uint32_t /*n*/, uint32_t i, uint32_t j) const {
    const int ci = s[i], cj = s[j];
    return t.init_multibranch_base +
           t.terminal_mismatch_scores_multibranch[cj][ci][s[j - 1]][s[i + 1]] + pen(ci, cj);
  }

  // get_accessible_score, src/utils.rs:384-411, uses_sentinel_bases = false
  __device__ float accessible(const uint8_t* s, uint32_t n, uint32_t i, uint32_t j) const {
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

  __device__ __forceinline__ float junction_single(const uint8_t* s, uint32_t y0,
                                                   uint32_t y1) const {
    const int a0 = s[y0], a1 = s[y1];
    return f.helix_close_scores[a0][a1] + f.terminal_mismatch_scores[a0][a1][s[y0 + 1]][s[y1 - 1]];
  }

  __device__ __forceinline__ float junction(const uint8_t* s, uint32_t n, uint32_t p0,
                                            uint32_t p1) const {
    const int b0 = s[p0], b1 = s[p1];
    return f.helix_close_scores[b0][b1] +
           (p0 < n - 1 ? f.dangling_scores_left[b0][b1][s[p0 + 1]] : 0.f) +
           (p1 > 0 ? f.dangling_scores_right[b0][b1][s[p1 - 1]] : 0.f);
  }

  __device__ float hairpin(const uint8_t* s, uint32_t /*n*/, uint32_t i, uint32_t j) const {
    uint32_t len = j - i - 1;
    if (len > RNAMC_MAX_LOOP_LEN) len = RNAMC_MAX_LOOP_LEN;
    return f.hairpin_scores_len_cumulative[len] + junction_single(s, i, j);
  }

  __device__ float twoloop(const uint8_t* s, uint32_t i, uint32_t j, uint32_t k, uint32_t l,
                           uint32_t a, uint32_t b) const {
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

  __device__ float mbclose(const uint8_t* s, uint32_t n, uint32_t i, uint32_t j) const {
    return f.multibranch_score_base + f.multibranch_score_basepair + junction(s, n, i, j);
  }

  __device__ float accessible(const uint8_t* s, uint32_t n, uint32_t i, uint32_t j) const {
    return junction(s, n, j, i) + f.basepair_scores[s[i]][s[j]];
  }
};

template <bool CONTRA>
struct ModelOf;
template <>
struct ModelOf<false> {
  using type = Turner;
  static __device__ __forceinline__ Turner make(const DeviceBatch& b) {
    return Turner{b.params->turner, b.hp_init};
  }
};
template <>
struct ModelOf<true> {
  using type = Contra;
  static __device__ __forceinline__ Contra make(const DeviceBatch& b) {
    return Contra{b.params->contra};
  }
};

// ----------------------------------------------------------------------------
// workspace initialisation: FoldSums::new (src/mccaskill_algo.rs:213-226) and
// the two outside matrices (528-529): sums_external = 0, everything else -inf.
__global__ void k_init(DeviceBatch b) {
  const SeqDesc sd = b.seqs[blockIdx.y];
  float* base = b.workspace + sd.ws_off;
  const size_t tri = sd.tri_pad;
  const size_t total = tri * M_COUNT;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t x = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; x < total;
       x += stride) {
    const size_t mat = x / tri;
    base[x] = (mat == M_Z) ? 0.f : kNegInf;
  }
  float* out = b.out + sd.out_off;
  const size_t olen = static_cast<size_t>(sd.n) * (sd.n + 1u) / 2u;
  for (size_t x = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; x < olen; x += stride)
    out[x] = kNegInf;
}

// ----------------------------------------------------------------------------
// inside pass, closing-pair block of diagonal d
// (src/mccaskill_algo.rs:297-343 Turner, 400-467 CONTRAfold)
template <bool CONTRA>
__global__ void k_inside_pair(DeviceBatch b, uint32_t d) {
  const Seq q = load_seq(b, blockIdx.y);
  const uint32_t n = q.n;
  if (d >= n) return;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - d) return;
  const uint32_t j = i + d;
  const uint8_t* s = q.s;
  if (!canonical(s[i], s[j])) return;
  if (!(b.allows_short_hairpins && CONTRA) && d + 1 < RNAMC_MIN_SPAN_HAIRPIN_CLOSE) return;
  const auto model = ModelOf<CONTRA>::make(b);

  float sum = kNegInf;
  if (!CONTRA || d - 1 <= RNAMC_MAX_LOOP_LEN) sum = lse(sum, model.hairpin(s, n, i, j));
  // enclosed pairs (k,l) = (i+1+a, j-1-b), a ascending, b ascending (l descending),
  // a+b <= 30, k < j-1, l > k   <=>   a + b <= d-3   (uniform over the diagonal)
  if (d >= 3) {
    const uint32_t lim = min(static_cast<uint32_t>(RNAMC_MAX_2LOOP_LEN), d - 3);
    const float* qb = q.m[M_QB];
    for (uint32_t a = 0; a <= lim; a++) {
      const uint32_t k = i + 1 + a;
      for (uint32_t bb = 0; bb <= lim - a; bb++) {
        const uint32_t l = j - 1 - bb;
        const float x = qb[tri_off(n, l - k) + k];
        if (x > kNegInf) {
          const float y = model.twoloop(s, i, j, k, l, a, bb);
          sum = lse(sum, x + y);
        }
      }
    }
  }
  const float mbc = model.mbclose(s, n, i, j);
  const float qm = (d >= 2) ? q.m[M_QM][tri_off(n, d - 2) + i + 1] : kNegInf;
  sum = lse(sum, qm + mbc);
  const float acc = model.accessible(s, n, i, j);
  if (sum > kNegInf) {
    const uint32_t o = tri_off(n, d) + i;
    q.m[M_MBC][o] = mbc;
    q.m[M_QB][o] = sum;
    q.m[M_QA][o] = sum + acc;
  }
}

// ----------------------------------------------------------------------------
// inside pass, the three Theta(n) folds of diagonal d
// (src/mccaskill_algo.rs:344-374 Turner, 468-512 CONTRAfold)
template <bool CONTRA>
__global__ void k_inside_sums(DeviceBatch b, uint32_t d) {
  const Seq q = load_seq(b, blockIdx.y);
  const uint32_t n = q.n;
  if (d >= n) return;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - d) return;
  const uint32_t od = tri_off(n, d) + i;  // this cell, diag-major
  const float* zre = q.m[M_ZRE];
  const float* zrm = q.m[CONTRA ? M_ZRM : M_ZRE];
  const float* z = q.m[M_Z];
  const float* q1 = q.m[M_Q1D];

  float zr_ext, zr_mb;
  if (!CONTRA) {
    // sums_rightmost_basepairs_external(i,j) = fold_{k=i+1..j} sums_accessible(i,k); the
    // terms do not depend on j, so the fold of (i,j-1) extended by one step is the
    // same sequence of operations (344-351).
    const float prev = (d >= 1) ? zre[tri_off(n, d - 1) + i] : kNegInf;
    zr_ext = lse(prev, q.m[M_QA][od]);
    zr_mb = zr_ext;
    q.m[M_ZRE][od] = zr_ext;
  } else {
    const rnamc_fold_score_sets& f = b.params->contra;
    const float ebp = f.external_score_basepair, eun = f.external_score_unpair;
    const float mbp = f.multibranch_score_basepair, mun = f.multibranch_score_unpair;
    const float* qa = q.m[M_QA];
    zr_ext = kNegInf;
    zr_mb = kNegInf;
    for (uint32_t t = 1; t <= d; t++) {  // k = i + t, j - k = d - t
      const float x = qa[tri_off(n, t) + i];
      if (x > kNegInf) {
        const float cnt = static_cast<float>(d - t);
        zr_ext = lse(zr_ext, x + ebp + eun * cnt);
        zr_mb = lse(zr_mb, x + mbp + mun * cnt);
      }
    }
    q.m[M_ZRE][od] = zr_ext;
    q.m[M_ZRM][od] = zr_mb;
  }

  // sums_external (352-363 / 487-498) and sums_1ormore / sums_multibranch
  // (364-374 / 499-512) share the operand Zr[k][j]; walk k = i + t once.
  float ext, s1, s2 = kNegInf;
  if (!CONTRA) {
    const float c = b.params->turner.coeff_num_branches;
    ext = lse(0.f, zr_ext + 0.f);  // k = i: Z[i][i-1] is the lower-triangle 0
    s1 = zr_ext + c;
    for (uint32_t t = 1; t < d; t++) {
      const float r = zre[tri_off(n, d - t) + i + t];
      const uint32_t o = tri_off(n, t - 1) + i;
      ext = lse(ext, r + z[o]);
      const float x = r + c;
      s1 = lse(s1, x);
      s2 = lse(s2, q1[o] + x);
    }
  } else {
    const rnamc_fold_score_sets& f = b.params->contra;
    const float mun = f.multibranch_score_unpair;
    ext = lse(f.external_score_unpair * static_cast<float>(d + 1), zr_ext + 0.f);
    s1 = zr_mb;
    for (uint32_t t = 1; t < d; t++) {
      const uint32_t or_ = tri_off(n, d - t) + i + t;
      const uint32_t o = tri_off(n, t - 1) + i;
      ext = lse(ext, zre[or_] + z[o]);
      const float x = zrm[or_];
      s1 = lse(s1, x + mun * static_cast<float>(t));
      s2 = lse(s2, q1[o] + x);
    }
  }
  if (d == 0) ext = CONTRA ? b.params->contra.external_score_unpair * 1.f : 0.f;
  q.m[M_Z][od] = ext;
  q.m[M_QM][od] = s2;
  s1 = lse(s1, s2);
  q.m[M_Q1D][od] = s1;
  q.m[M_Q1R][tri_off(n, i) + d] = s1;
}

// ----------------------------------------------------------------------------
// outside pass of diagonal d (src/mccaskill_algo.rs:537-605 Turner, 638-718
// CONTRAfold).  basepair_probs stays in the log domain in `out` until k_finalize.
template <bool CONTRA>
__global__ void k_outside(DeviceBatch b, uint32_t d) {
  const Seq q = load_seq(b, blockIdx.y);
  const uint32_t n = q.n;
  if (d >= n) return;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - d) return;
  const uint32_t j = i + d;
  const uint8_t* s = q.s;
  const float* q1d = q.m[M_Q1D];
  const float* w = q.m[M_W];

  // probs_multibranch / probs_multibranch2: k = j + t over pairs (i,k)
  float pm = kNegInf, pm2 = kNegInf;
  {
    const float mun = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
    const uint32_t cnt = n - 1 - j;
    for (uint32_t t = 1; t <= cnt; t++) {
      const float x = w[tri_off(n, d + t) + i];  // -inf when (i,k) is no pair
      // sums_1ormore_basepairs[j+1][k-1]; for t == 1 that is the empty interval
      const float r = (t >= 2) ? q1d[tri_off(n, t - 2) + j + 1] : kNegInf;
      pm = lse(pm, x + r);
      if (CONTRA) {
        pm2 = lse(pm2, x + mun * static_cast<float>(t - 1));
      } else {
        pm2 = lse(pm2, x);
      }
    }
  }
  const uint32_t orow = tri_off(n, i) + d;
  q.m[M_PM][orow] = pm;
  q.m[M_PM2][orow] = pm2;

  const uint32_t od = tri_off(n, d) + i;
  const float qb_ij = q.m[M_QB][od];
  if (!(qb_ij > kNegInf)) return;
  const auto model = ModelOf<CONTRA>::make(b);
  const float qa_ij = q.m[M_QA][od];
  const float* z = q.m[M_Z];
  const float ztot = z[tri_off(n, n - 1)];
  const float zl = (i < 1) ? 0.f : z[tri_off(n, i - 1)];                           // Z[0][i-1]
  const float zr = (j > n - 2) ? 0.f : z[tri_off(n, n - 2 - j) + j + 1];           // Z[j+1][n-1]
  float p;
  if (CONTRA) {
    p = zl + zr + qa_ij + b.params->contra.external_score_basepair - ztot;
  } else {
    p = zl + qa_ij + zr - ztot;
  }
  // enclosing pairs (k,l) = (i-1-a, j+1+bb), a ascending (k descending), bb ascending,
  // a + bb <= 30
  {
    const float* qb = q.m[M_QB];
    const float* lp = q.out;
    const uint32_t amax = min(static_cast<uint32_t>(RNAMC_MAX_2LOOP_LEN), i == 0 ? 0u : i - 1);
    if (i > 0) {
      for (uint32_t a = 0; a <= amax; a++) {
        const uint32_t k = i - 1 - a;
        for (uint32_t bb = 0; bb <= RNAMC_MAX_2LOOP_LEN - a; bb++) {
          const uint32_t l = j + 1 + bb;
          if (l >= n) break;
          const uint32_t o = tri_off(n, l - k) + k;
          const float x = qb[o];
          if (x > kNegInf) {
            const float y = model.twoloop(s, k, l, i, j, a, bb);
            p = lse(p, lp[o] + qb_ij - x + y);
          }
        }
      }
    }
  }
  // multibranch contexts: k = 0..i-1 closes nothing here; (k, l>j) pairs were
  // folded into probs_multibranch{,2}[k][j]
  {
    const float* q1r = q.m[M_Q1R];
    const float* pmr = q.m[M_PM];
    const float* pm2r = q.m[M_PM2];
    float sa;
    float mun = 0.f;
    if (CONTRA) {
      sa = qa_ij + b.params->contra.multibranch_score_basepair;
      mun = b.params->contra.multibranch_score_unpair;
    } else {
      sa = qa_ij + b.params->turner.coeff_num_branches;
    }
    for (uint32_t k = 0; k < i; k++) {
      const uint32_t rk = tri_off(n, k);
      // sums_1ormore_basepairs[k+1][i-1]; empty interval when k+1 > i-1
      const float x = (k + 2 <= i) ? q1r[tri_off(n, k + 1) + (i - 1) - (k + 1)] : kNegInf;
      const float y2 = pm2r[rk + j - k];
      const float y = pmr[rk + j - k];
      p = lse(p, sa + y2 + x);
      if (CONTRA) {
        p = lse(p, sa + y + mun * static_cast<float>(i - k - 1));
      } else {
        p = lse(p, sa + y);
      }
      p = lse(p, sa + x + y);
    }
  }
  if (p > kNegInf) {
    q.out[od] = p;
    q.m[M_W][od] = p + q.m[M_MBC][od] - qb_ij;
  }
}

// final map (src/mccaskill_algo.rs:608 / 721) + log partition function
__global__ void k_finalize(DeviceBatch b) {
  const SeqDesc sd = b.seqs[blockIdx.y];
  float* out = b.out + sd.out_off;
  const size_t olen = static_cast<size_t>(sd.n) * (sd.n + 1u) / 2u;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t x = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; x < olen;
       x += stride) {
    const float lp = out[x];
    out[x] = (lp > kNegInf) ? expf_ref(lp) : -1.0f;
  }
  if (b.log_partition && blockIdx.x == 0 && threadIdx.x == 0) {
    const float* z = b.workspace + sd.ws_off + static_cast<size_t>(M_Z) * sd.tri_pad;
    b.log_partition[sd.batch_idx] = z[tri_off(sd.n, sd.n - 1)];
  }
}

}  // namespace

// ----------------------------------------------------------------------------
// launch wrappers (host)

static inline dim3 diag_grid(uint32_t cells, uint32_t block, uint32_t nseq) {
  return dim3((cells + block - 1) / block, nseq, 1);
}

void launch_init(const DeviceBatch& b, uint32_t nseq, uint32_t max_n, hipStream_t st) {
  const uint64_t elems = static_cast<uint64_t>(max_n) * (max_n + 1) / 2 * M_COUNT;
  uint32_t gx = static_cast<uint32_t>(std::min<uint64_t>((elems + 1023) / 1024, 512));
  if (gx == 0) gx = 1;
  hipLaunchKernelGGL(k_init, dim3(gx, nseq, 1), dim3(256), 0, st, b);
}

void launch_inside(const DeviceBatch& b, bool contra, uint32_t d, uint32_t cells, uint32_t nseq,
                   uint32_t block, hipStream_t st) {
  const dim3 g = diag_grid(cells, block, nseq);
  if (contra) {
    hipLaunchKernelGGL(k_inside_pair<true>, g, dim3(block), 0, st, b, d);
    hipLaunchKernelGGL(k_inside_sums<true>, g, dim3(block), 0, st, b, d);
  } else {
    hipLaunchKernelGGL(k_inside_pair<false>, g, dim3(block), 0, st, b, d);
    hipLaunchKernelGGL(k_inside_sums<false>, g, dim3(block), 0, st, b, d);
  }
}

void launch_outside(const DeviceBatch& b, bool contra, uint32_t d, uint32_t cells, uint32_t nseq,
                    uint32_t block, hipStream_t st) {
  const dim3 g = diag_grid(cells, block, nseq);
  if (contra) {
    hipLaunchKernelGGL(k_outside<true>, g, dim3(block), 0, st, b, d);
  } else {
    hipLaunchKernelGGL(k_outside<false>, g, dim3(block), 0, st, b, d);
  }
}

void launch_finalize(const DeviceBatch& b, uint32_t nseq, uint32_t max_n, hipStream_t st) {
  const uint64_t elems = static_cast<uint64_t>(max_n) * (max_n + 1) / 2;
  uint32_t gx = static_cast<uint32_t>(std::min<uint64_t>((elems + 1023) / 1024, 256));
  if (gx == 0) gx = 1;
  hipLaunchKernelGGL(k_finalize, dim3(gx, nseq, 1), dim3(256), 0, st, b);
}

}  // namespace rnamc
