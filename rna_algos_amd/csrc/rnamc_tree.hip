// rnamc_tree.hip — tree-order ("fast") summation mode of the McCaskill inside/outside sweep
// (rnamc_ctx_set "summation_mode" 1).  Hand-written gfx950 kernels.
//
// What is computed: the recurrences of the reference (src/mccaskill_algo.rs:282-723 of
// heartsh/rna-algos; loops 344-374, 468-512, 540-557, 594-601; fold src/utils.rs:579-627) with
// every logsumexp fold evaluated as an ORDER-FREE sum: the terms of a cell's sum are spread
// over the 256 lanes of a workgroup, every lane keeps a running (max, sum of exp) pair,
// and the pairs are merged by DPP wave reductions and a 4-entry LDS exchange.  exp / log are
// the hardware's v_exp_f32 / v_log_f32 (1 ulp), not the reference's 8-piece cubics.
//
// This mode CANNOT be bit-compared with the reference: its left fold is approximate and
// non-associative.  It is validated against the f64 evaluation of the
// same recurrences (oracle/mccaskill_exact.c, oracle/bruteforce.c) and its deviation from the
// reference-order mode is measured and asserted in tests/test_gpu_tree.py.
//
// Being free of the summation order, the mode also drops the Theta(n^3) loops whose terms
// do not depend on the cell:
//   * sums_rightmost_basepairs_* (344-351, 468-486) are one step per cell:
//       Zr(i,j) = (Zr(i,j-1) + unpair) (+) (Qa(i,j) + basepair);
//   * the first fold of L_c (364-374 `sum`, 499-512) is a column prefix:
//       U(i,j) = (U(i+1,j) + unpair) (+) Zr_mb(i,j);       sums_1ormore = U (+) sums_multibranch
//   * probs_multibranch2 (548-556, 654-657) is a row prefix, and the second case of L_e
//     (596-600, 707-712) a column prefix SP of probs_multibranch; cases one and three of L_e
//     share their operand: Q1(k+1,i-1) + [Pm2(k,j) (+) Pm(k,j)] =: Q1 + R(k,j);
//   * sums_external is needed only as Z[0][i-1] and Z[j+1][n-1] (561-573, 676-680): the prefix
//     row Zp(j) = Z(0,j) keeps the reference's rightmost-pair decomposition (352-363), the
//     suffix column Zs(i) = Z(i,n-1) uses the mirror (leftmost-pair) one,
//       Zs(i) = (Zs(i+1) + unpair) (+) (+)_l (Qa(i,l) + basepair + Zs(l+1)),
//     the same set of structures with the same weights.
// What remains cubic are the three (logsumexp,+) products sums_multibranch (L_c), probs_multibranch
// (L_d) and the Q1 x R part of L_e.  Their operands are stored so that the k index is
// contiguous for both factors (row-major x column-major), one workgroup per cell streams them
// coalesced, 4 B per lane per load.
//
// One launch per anti-diagonal and pass; blockIdx.x = cell, blockIdx.y = sequence.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "rnamc_device.h"
#include "rnamc_scoring.h"

namespace rnamc {

namespace {

constexpr float kL2E = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kEmpty = -1.0e30f;  // running max of an accumulator without terms
constexpr int kThreads = 256;

__device__ __forceinline__ float vmaxf(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }

// Sum of exp(x_t) held as s * exp(m), m >= every x_t seen (m = kEmpty, s = 0 when empty).
// After any finite term s >= 1, so log2(s) never meets a denormal.
struct Acc {
  float m, s;
};
__device__ __forceinline__ Acc acc_empty() { return Acc{kEmpty, 0.f}; }
__device__ __forceinline__ void acc_add(Acc& a, float x) {
  const float mn = vmaxf(a.m, x);
  a.s = __builtin_fmaf(a.s, ex2((a.m - mn) * kL2E), ex2((x - mn) * kL2E));
  a.m = mn;
}
__device__ __forceinline__ void acc_add4(Acc& a, float x0, float x1, float x2, float x3) {
  const float mn = vmaxf(vmaxf(a.m, vmaxf(x0, x1)), vmaxf(x2, x3));
  const float mb = mn * kL2E;
  const float e01 = ex2(__builtin_fmaf(x0, kL2E, -mb)) + ex2(__builtin_fmaf(x1, kL2E, -mb));
  const float e23 = ex2(__builtin_fmaf(x2, kL2E, -mb)) + ex2(__builtin_fmaf(x3, kL2E, -mb));
  a.s = __builtin_fmaf(a.s, ex2((a.m - mn) * kL2E), e01 + e23);
  a.m = mn;
}
__device__ __forceinline__ void acc_merge(Acc& a, const Acc& b) {
  const float mn = vmaxf(a.m, b.m);
  a.s = a.s * ex2((a.m - mn) * kL2E) + b.s * ex2((b.m - mn) * kL2E);
  a.m = mn;
}
__device__ __forceinline__ float acc_value(const Acc& a) {
  return a.s > 0.f ? __builtin_fmaf(lg2(a.s), kLn2, a.m) : kNegInf;
}
// exact two-term logsumexp of finite-or--inf operands
__device__ __forceinline__ float lse2(float a, float b) {
  const float hi = vmaxf(a, b);
  const float lo = fminf(a, b);
  if (!(lo > kNegInf)) return hi;
  return __builtin_fmaf(lg2(1.f + ex2((lo - hi) * kL2E)), kLn2, hi);
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp(float old, float x) {
  return __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_update_dpp(
      static_cast<int>(__float_as_uint(old)), static_cast<int>(__float_as_uint(x)), CTRL, ROWMASK,
      0xF, false)));
}
// inclusive scan inside the rows of 16 (row_shr 1,2,4,8), then row_bcast:15 into rows 1 and 3
// and row_bcast:31 into rows 2 and 3: lane 63 holds the reduction of the wave
__device__ __forceinline__ float wave_max(float v) {
  v = vmaxf(v, dpp<0x111, 0xF>(kNegInf, v));
  v = vmaxf(v, dpp<0x112, 0xF>(kNegInf, v));
  v = vmaxf(v, dpp<0x114, 0xF>(kNegInf, v));
  v = vmaxf(v, dpp<0x118, 0xF>(kNegInf, v));
  v = vmaxf(v, dpp<0x142, 0xA>(kNegInf, v));
  v = vmaxf(v, dpp<0x143, 0xC>(kNegInf, v));
  return __uint_as_float(static_cast<uint32_t>(
      __builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(v)), 63)));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp<0x111, 0xF>(0.f, v);
  v += dpp<0x112, 0xF>(0.f, v);
  v += dpp<0x114, 0xF>(0.f, v);
  v += dpp<0x118, 0xF>(0.f, v);
  v += dpp<0x142, 0xA>(0.f, v);
  v += dpp<0x143, 0xC>(0.f, v);
  return __uint_as_float(static_cast<uint32_t>(
      __builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(v)), 63)));
}
__device__ __forceinline__ Acc wave_reduce(const Acc& a) {
  const float m = wave_max(a.m);
  const float s = wave_sum(a.s * ex2((a.m - m) * kL2E));
  return Acc{m, s};
}

// Workgroup reduction of NA accumulators: every wave reduces its own, lane 0 parks the pair
// in LDS, and after the barrier every thread merges the four pairs (so that each thread holds
// the workgroup's totals without a second barrier).
template <int NA>
__device__ __forceinline__ void block_reduce(Acc (&a)[NA], float (*lds)[NA][2]) {
  const uint32_t wave = threadIdx.x >> 6;
#pragma unroll
  for (int x = 0; x < NA; x++) {
    const Acc r = wave_reduce(a[x]);
    if ((threadIdx.x & 63u) == 0u) {
      lds[wave][x][0] = r.m;
      lds[wave][x][1] = r.s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int x = 0; x < NA; x++) {
    Acc t = Acc{lds[0][x][0], lds[0][x][1]};
#pragma unroll
    for (int w = 1; w < kThreads / 64; w++) acc_merge(t, Acc{lds[w][x][0], lds[w][x][1]});
    a[x] = t;
  }
}

// (+)_k (A[k] + B[k]) over k in [0, len): both operands contiguous in k, the workgroup's
// lanes take consecutive k (256-B wave accesses), four loads of each operand in flight.
__device__ __forceinline__ void acc_product(Acc& a, const float* __restrict__ A,
                                            const float* __restrict__ B, uint32_t len) {
  for (uint32_t k = threadIdx.x; k < len; k += 4u * kThreads) {
    const uint32_t k1 = k + kThreads, k2 = k + 2u * kThreads, k3 = k + 3u * kThreads;
    const float a0 = A[k], b0 = B[k];
    const float a1 = k1 < len ? A[k1] : kNegInf, b1 = k1 < len ? B[k1] : kNegInf;
    const float a2 = k2 < len ? A[k2] : kNegInf, b2 = k2 < len ? B[k2] : kNegInf;
    const float a3 = k3 < len ? A[k3] : kNegInf, b3 = k3 < len ? B[k3] : kNegInf;
    acc_add4(a, a0 + b0, a1 + b1, a2 + b2, a3 + b3);
  }
}

struct TSeq {
  const uint8_t* s;
  uint32_t n, ld;
  float* m[T_COUNT];
  float* zp;   // zp[x] = Z(0, x-1), zp[0] = 0            (n + 1 entries)
  float* zs;   // zs[x] = Z(x, n-1), zs[n] = 0            (n + 1 entries)
  float* out;  // packed diagonal-major triangle: log bpp until k_tree_finalize
  uint32_t batch_idx;
};

__device__ __forceinline__ TSeq load_tseq(const TreeBatch& b, uint32_t which) {
  const TreeSeq sd = b.use_one ? b.one : b.seqs[which];
  TSeq q;
  q.s = b.bases + sd.seq_off;
  q.n = sd.n;
  q.ld = sd.ld;
  float* base = b.workspace + sd.ws_off;
#pragma unroll
  for (int x = 0; x < T_COUNT; x++) q.m[x] = base + static_cast<size_t>(x) * sd.msz;
  q.zp = base + static_cast<size_t>(T_COUNT) * sd.msz;
  q.zs = q.zp + (sd.n + 64u);
  q.out = b.out + sd.out_off;
  q.batch_idx = sd.batch_idx;
  return q;
}

__device__ __forceinline__ uint32_t tri_off(uint32_t n, uint32_t d) {
  return d * n - (d * (d - 1u)) / 2u;
}

// The <= 496 (a, b) pairs with a + b <= 30 (src/mccaskill_algo.rs:306-315) in 512 slots: slot
// row r < 15 holds the 31 - r pairs of a = r followed by the r + 1 pairs of a = 30 - r; row 15
// the 16 pairs of a = 15.
__device__ __forceinline__ bool probe_slot(uint32_t p, uint32_t& a, uint32_t& bb) {
  const uint32_t r = p >> 5, c = p & 31u;
  const bool first = c < 31u - r;
  a = first ? r : 30u - r;
  bb = first ? c : c - (31u - r);
  return r < 15u || c < 16u;
}
static_assert(RNAMC_MAX_2LOOP_LEN == 30 && RNAMC_MAX_LOOP_LEN == 30, "probe_slot covers a + b <= 30");

template <bool CONTRA>
struct TModel;
template <>
struct TModel<false> {
  static __device__ __forceinline__ Turner make(const TreeBatch& b) {
    return Turner{b.params->turner, b.hp_init};
  }
};
template <>
struct TModel<true> {
  static __device__ __forceinline__ Contra make(const TreeBatch& b) { return Contra{b.params->contra}; }
};

// ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_tree_init(TreeBatch b, int contra, int what) {
  const TreeSeq sd = b.use_one ? b.one : b.seqs[blockIdx.y];
  float* base = b.workspace + sd.ws_off;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  const size_t t0 = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (what == 0) {
    // FoldSums::new (src/mccaskill_algo.rs:213-226): every sparse / dense sum starts absent
    const size_t total = static_cast<size_t>(T_COUNT) * sd.msz;
    for (size_t x = t0; x < total; x += stride) base[x] = kNegInf;
    float* out = b.out + sd.out_off;
    const size_t olen = static_cast<size_t>(sd.n) * (sd.n + 1u) / 2u;
    for (size_t x = t0; x < olen; x += stride) out[x] = kNegInf;
    // sums_external of the spans nothing can pair in: the unpaired structure alone
    // (0 under Turner: 357-363 with no term; external_score_unpair * len: 487)
    float* zp = base + static_cast<size_t>(T_COUNT) * sd.msz;
    float* zs = zp + (sd.n + 64u);
    const float unp = contra ? b.params->contra.external_score_unpair : 0.f;
    for (size_t x = t0; x <= sd.n; x += stride) {
      zp[x] = contra ? unp * static_cast<float>(x) : 0.f;
      zs[x] = contra ? unp * static_cast<float>(sd.n - x) : 0.f;
    }
  } else {
    // the four slots the outside sweep reuses (W, R, Pm2, SP)
    const int mats[4] = {T_ZRE, T_ZRM, T_QM, T_U};
    for (int y = 0; y < 4; y++) {
      float* p = base + static_cast<size_t>(mats[y]) * sd.msz;
      for (size_t x = t0; x < sd.msz; x += stride) p[x] = kNegInf;
    }
  }
}

// ----------------------------------------------------------------------------
// inside pass, one workgroup per cell (i, i+d)
template <bool CONTRA>
__global__ void __launch_bounds__(256) k_tree_inside(TreeBatch b, uint32_t d) {
  __shared__ float red[kThreads / 64][4][2];
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  const uint32_t i = blockIdx.x;
  if (i + d >= n) return;
  const uint32_t j = i + d;
  const uint32_t t = threadIdx.x;
  const uint8_t* __restrict__ s = q.s;
  const auto model = TModel<CONTRA>::make(b);
  const float* __restrict__ qb_r = q.m[T_QB];
  const size_t row_i = static_cast<size_t>(i) * ld, col_j = static_cast<size_t>(j) * ld;

  bool act = canonical(s[i], s[j]);
  if (!(b.allows_short_hairpins && CONTRA) && d + 1 < RNAMC_MIN_SPAN_HAIRPIN_CLOSE) act = false;

  // thread 0 owns the cell's scalar recurrences: fetch their operands first
  float zr_e_prev = kNegInf, zr_m_prev = kNegInf, u_next = kNegInf;
  if (t == 0) {
    if (j >= 1) {
      zr_e_prev = q.m[T_ZRE][col_j - ld + i];
      if (CONTRA) zr_m_prev = q.m[T_ZRM][col_j - ld + i];
    }
    u_next = q.m[T_U][col_j + i + 1];  // (i+1 == n: the column's pad, -inf)
  }

  Acc acc[4] = {acc_empty(), acc_empty(), acc_empty(), acc_empty()};
  // [0] closing-pair block (297-343 / 400-467)
  if (act) {
    if (t == 0 && (!CONTRA || d - 1 <= RNAMC_MAX_LOOP_LEN)) acc_add(acc[0], model.hairpin(s, n, i, j));
    if (t == 1 && d >= 2)
      acc_add(acc[0], q.m[T_QM][row_i + ld + (j - 1)] + model.mbclose(s, n, i, j));
    if (d >= 3) {
#pragma unroll
      for (uint32_t p = t; p < 512u; p += kThreads) {
        uint32_t a, bb;
        if (probe_slot(p, a, bb) && a + bb + 3u <= d) {
          const uint32_t k = i + 1u + a, l = j - 1u - bb;
          const float x = qb_r[static_cast<size_t>(k) * ld + l];
          if (x > kNegInf) acc_add(acc[0], x + model.twoloop(s, i, j, k, l));
        }
      }
    }
  }
  // [1] sums_multibranch (L_c second fold): k = i+1 .. j-1, Q1(i,k-1) + Zr_mb(k,j)
  if (d >= 2) acc_product(acc[1], q.m[T_Q1R] + row_i + i, q.m[T_ZRM] + col_j + i + 1, d - 1);
  // [2] Z(0,j): k = 1 .. j, Zr_ext(k,j) + Z(0,k-1)       (the k = 0 term is this cell's own)
  if (i == 0 && j >= 1) acc_product(acc[2], q.m[T_ZRE] + col_j + 1, q.zp + 1, j);
  // [3] Z(i,n-1): l = i+1 .. n-2, Qa(i,l) + Z(l+1,n-1)   (l = n-1 is this cell's own)
  if (j == n - 1 && d >= 1) acc_product(acc[3], q.m[T_QA] + row_i + i + 1, q.zs + i + 2, d - 1);
  block_reduce<4>(acc, red);
  if (t != 0) return;

  const float ext_bp = CONTRA ? b.params->contra.external_score_basepair : 0.f;
  const float ext_un = CONTRA ? b.params->contra.external_score_unpair : 0.f;
  const float mb_bp = CONTRA ? b.params->contra.multibranch_score_basepair
                             : b.params->turner.coeff_num_branches;
  const float mb_un = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  float qa = kNegInf;
  if (act) {
    const float qb = acc_value(acc[0]);
    if (qb > kNegInf) {
      qa = qb + model.accessible(s, n, i, j);
      q.m[T_QB][row_i + j] = qb;
      q.m[T_QA][row_i + j] = qa;
    }
  }
  // sums_rightmost_basepairs_{external,multibranch}: one step from the cell to the left
  const float zr_e = lse2(zr_e_prev + ext_un, qa + ext_bp);
  const float zr_m = CONTRA ? lse2(zr_m_prev + mb_un, qa + mb_bp) : zr_e + mb_bp;
  q.m[T_ZRE][col_j + i] = zr_e;
  q.m[T_ZRM][col_j + i] = zr_m;
  const float u = lse2(u_next + mb_un, zr_m);
  q.m[T_U][col_j + i] = u;
  const float qm = acc_value(acc[1]);
  q.m[T_QM][row_i + j] = qm;
  const float q1 = lse2(u, qm);
  q.m[T_Q1R][row_i + j] = q1;
  q.m[T_Q1C][col_j + i] = q1;
  if (i == 0) {
    // sums_external[0][j] (352-363 / 487-498)
    Acc z = acc[2];
    acc_add(z, zr_e);  // k = 0: Z(0,-1) = 0
    acc_add(z, CONTRA ? ext_un * static_cast<float>(j + 1) : 0.f);
    q.zp[j + 1] = acc_value(z);
  }
  if (j == n - 1) {
    Acc z = acc[3];
    z.m += ext_bp;            // every product term carries the pair's external_score_basepair
    acc_add(z, qa + ext_bp);  // l = n-1: Z(n,n-1) = 0
    acc_add(z, q.zs[i + 1] + ext_un);
    q.zs[i] = acc_value(z);
  }
}

// ----------------------------------------------------------------------------
// outside pass, one workgroup per cell (i, i+d)
template <bool CONTRA>
__global__ void __launch_bounds__(256) k_tree_outside(TreeBatch b, uint32_t d) {
  __shared__ float red[kThreads / 64][3][2];
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  const uint32_t i = blockIdx.x;
  if (i + d >= n) return;
  const uint32_t j = i + d;
  const uint32_t t = threadIdx.x;
  const uint8_t* __restrict__ s = q.s;
  const auto model = TModel<CONTRA>::make(b);
  const float* __restrict__ qb_r = q.m[T_QB];
  const size_t row_i = static_cast<size_t>(i) * ld, col_j = static_cast<size_t>(j) * ld;
  const float* __restrict__ w_r = q.m[T_ZRE];   // W = (P + mbclose) - Qb, row-major
  float* __restrict__ r_c = q.m[T_ZRM];         // R = Pm (+) Pm2, column-major
  float* __restrict__ pm2_r = q.m[T_QM];        // probs_multibranch2, row-major
  float* __restrict__ sp_c = q.m[T_U];          // sp_c(i,j) = (+)_{k<=i} Pm(k,j) [+ unpaired], column-major

  const float qb = qb_r[row_i + j];
  const bool paired = qb > kNegInf;  // (uniform over the workgroup)
  float pm2_next = kNegInf, w_next = kNegInf, sp_prev = kNegInf;
  if (t == 0) {
    if (j + 1 < n) {
      pm2_next = pm2_r[row_i + j + 1];
      w_next = w_r[row_i + j + 1];
    }
    if (i >= 1) sp_prev = sp_c[col_j + i - 1];
  }
  Acc acc[3] = {acc_empty(), acc_empty(), acc_empty()};
  // [0] probs_multibranch (L_d): k = j+1 .. n-1, W(i,k) + Q1(j+1,k-1)
  if (j + 1 < n)
    acc_product(acc[0], w_r + row_i + j + 1, q.m[T_Q1R] + static_cast<size_t>(j + 1) * ld + j, n - 1 - j);
  if (paired) {
    // [1] enclosing 2-loops (574-593 / 681-700): (k,l) = (i-1-a, j+1+b)
#pragma unroll
    for (uint32_t p = t; p < 512u; p += kThreads) {
      uint32_t a, bb;
      if (probe_slot(p, a, bb) && a < i && j + 1u + bb < n) {
        const uint32_t k = i - 1u - a, l = j + 1u + bb;
        const float x = qb_r[static_cast<size_t>(k) * ld + l];
        if (x > kNegInf) {
          const float pkl = q.out[tri_off(n, l - k) + k];
          acc_add(acc[1], ((pkl + qb) - x) + model.twoloop(s, k, l, i, j));
        }
      }
    }
    // [2] L_e cases one and three: k = 0 .. i-1, Q1(k+1,i-1) + R(k,j)
    if (i >= 1) acc_product(acc[2], q.m[T_Q1C] + static_cast<size_t>(i - 1) * ld + 1, r_c + col_j, i);
  }
  block_reduce<3>(acc, red);
  if (t != 0) return;

  const float mb_un = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  const float pm = acc_value(acc[0]);
  const float pm2 = lse2(pm2_next + mb_un, w_next);
  pm2_r[row_i + j] = pm2;
  r_c[col_j + i] = lse2(pm, pm2);
  sp_c[col_j + i] = lse2(sp_prev + mb_un, pm);
  if (!paired) return;
  const float qa = q.m[T_QA][row_i + j];
  const float ztot = q.zp[n];
  // exterior term (561-573 / 676-680)
  float ext = CONTRA ? (((q.zp[i] + q.zs[j + 1]) + qa) + b.params->contra.external_score_basepair) - ztot
                     : ((q.zp[i] + qa) + q.zs[j + 1]) - ztot;
  Acc pa = acc[1];
  acc_add(pa, ext);
  // L_e: every term carries A = Qa + (coeff_num_branches | multibranch_score_basepair)
  const float A = qa + (CONTRA ? b.params->contra.multibranch_score_basepair
                               : b.params->turner.coeff_num_branches);
  acc_add(pa, A + acc_value(acc[2]));
  acc_add(pa, A + sp_prev);  // case two: sp_prev holds the unpaired factors of rows k < i already
  const float lp = acc_value(pa);
  if (lp > kNegInf) {
    q.out[tri_off(n, d) + i] = lp;
    q.m[T_ZRE][row_i + j] = (lp + model.mbclose(s, n, i, j)) - qb;
  }
}

__global__ void __launch_bounds__(256) k_tree_finalize(TreeBatch b) {
  const TreeSeq sd = b.use_one ? b.one : b.seqs[blockIdx.y];
  float* out = b.out + sd.out_off;
  const size_t olen = static_cast<size_t>(sd.n) * (sd.n + 1u) / 2u;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t x = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; x < olen; x += stride) {
    const float lp = out[x];
    out[x] = lp > kNegInf ? expf(lp) : -1.0f;
  }
  if (b.log_partition && blockIdx.x == 0 && threadIdx.x == 0) {
    const float* zp = b.workspace + sd.ws_off + static_cast<size_t>(T_COUNT) * sd.msz;
    b.log_partition[sd.batch_idx] = zp[sd.n];  // sums_external[0][n-1]
  }
}

}  // namespace

void launch_tree_init(const TreeBatch& b, uint32_t nseq, uint32_t max_n, bool contra, int what,
                      hipStream_t st) {
  const uint64_t elems = static_cast<uint64_t>(max_n) * max_n * (what == 0 ? T_COUNT : 4);
  uint32_t gx = static_cast<uint32_t>(std::min<uint64_t>((elems + 2047) / 2048, 2048));
  if (gx == 0) gx = 1;
  hipLaunchKernelGGL(k_tree_init, dim3(gx, nseq, 1), dim3(256), 0, st, b, contra ? 1 : 0, what);
}

void launch_tree_inside(const TreeBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                        hipStream_t st) {
  const dim3 grid(max_n - d, nseq, 1);
  if (contra)
    hipLaunchKernelGGL(k_tree_inside<true>, grid, dim3(kThreads), 0, st, b, d);
  else
    hipLaunchKernelGGL(k_tree_inside<false>, grid, dim3(kThreads), 0, st, b, d);
}

void launch_tree_outside(const TreeBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                         hipStream_t st) {
  const dim3 grid(max_n - d, nseq, 1);
  if (contra)
    hipLaunchKernelGGL(k_tree_outside<true>, grid, dim3(kThreads), 0, st, b, d);
  else
    hipLaunchKernelGGL(k_tree_outside<false>, grid, dim3(kThreads), 0, st, b, d);
}

void launch_tree_finalize(const TreeBatch& b, uint32_t nseq, uint32_t max_n, hipStream_t st) {
  const uint64_t elems = static_cast<uint64_t>(max_n) * (max_n + 1) / 2;
  uint32_t gx = static_cast<uint32_t>(std::min<uint64_t>((elems + 1023) / 1024, 1024));
  if (gx == 0) gx = 1;
  hipLaunchKernelGGL(k_tree_finalize, dim3(gx, nseq, 1), dim3(256), 0, st, b);
}

}  // namespace rnamc
