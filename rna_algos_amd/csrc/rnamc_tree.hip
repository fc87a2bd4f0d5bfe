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
__device__ __forceinline__ void acc_add2(Acc& a, float x0, float x1) {
  const float mn = vmaxf(a.m, vmaxf(x0, x1));
  const float mb = mn * kL2E;
  const float e01 = ex2(__builtin_fmaf(x0, kL2E, -mb)) + ex2(__builtin_fmaf(x1, kL2E, -mb));
  a.s = __builtin_fmaf(a.s, ex2((a.m - mn) * kL2E), e01);
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

// Reduction of NA accumulators over the TPC threads that share a cell.  TPC == 64: the wave's
// own DPP reduction, no LDS, no barrier.  Larger groups: every wave parks its pair in LDS,
// after the barrier the first wave reduces the W = TPC / 64 pairs once more; the other waves
// are done (returns false for them).
template <int NA, int TPC>
__device__ __forceinline__ bool cell_reduce(Acc (&a)[NA], float (*lds)[NA][2]) {
#pragma unroll
  for (int x = 0; x < NA; x++) a[x] = wave_reduce(a[x]);
  if (TPC == 64) return true;
  constexpr int W = TPC / 64;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (lane == 0u) {
#pragma unroll
    for (int x = 0; x < NA; x++) {
      lds[wave][x][0] = a[x].m;
      lds[wave][x][1] = a[x].s;
    }
  }
  __syncthreads();
  if (wave != 0u) return false;
#pragma unroll
  for (int x = 0; x < NA; x++) {
    Acc t = lane < static_cast<uint32_t>(W) ? Acc{lds[lane][x][0], lds[lane][x][1]} : acc_empty();
    a[x] = wave_reduce(t);
  }
  return true;
}

// (+)_k (A[k] + B[k]) over k in [0, len): both operands contiguous in k, the cell's TPC
// lanes take consecutive k (256-B wave accesses), four loads of each operand in flight.
template <int TPC>
__device__ __forceinline__ void acc_product(Acc& a, const float* __restrict__ A,
                                            const float* __restrict__ B, uint32_t len, uint32_t t) {
  for (uint32_t k = t; k < len; k += 4u * TPC) {
    const uint32_t k1 = k + TPC, k2 = k + 2u * TPC, k3 = k + 3u * TPC;
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
  const uint32_t* pk;  // 2-bit packed bases, 16 per word, position p at bit 2(p+32)
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
  q.pk = reinterpret_cast<const uint32_t*>(b.workspace + sd.pk_off);
  q.out = b.out + sd.out_off;
  q.batch_idx = sd.batch_idx;
  return q;
}

__device__ __forceinline__ uint32_t tri_off(uint32_t n, uint32_t d) {
  return d * n - (d * (d - 1u)) / 2u;
}

// 32 consecutive bases p0 .. p0+31 in one 64-bit value (window position q at bits 2q, 2q+1);
// p0 >= -32 (the packed copy carries 32 zero bases in front and >= 64 behind)
__device__ __forceinline__ uint64_t load_win64(const uint32_t* __restrict__ pk, int p0) {
  const uint32_t bit = 2u * static_cast<uint32_t>(p0 + 32);
  const uint32_t w = bit >> 5, sh = bit & 31u;
  const uint32_t w0 = pk[w], w1 = pk[w + 1], w2 = pk[w + 2];
  const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh);
  const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sh);
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
__device__ __forceinline__ int wb(uint64_t w, uint32_t q) { return static_cast<int>((w >> (2u * q)) & 3u); }

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
  static __device__ __forceinline__ float twoloop(const TreeBatch& b, uint32_t a, uint32_t bb, int ci,
                                                  int cj, int x1, int x2, int y1, int y2, int ak,
                                                  int al, int m2, int m3) {
    return turner_twoloop_flat(b.params->turner, a, bb, ci, cj, x1, x2, y1, y2, ak, al, m2, m3);
  }
};
template <>
struct TModel<true> {
  static __device__ __forceinline__ Contra make(const TreeBatch& b) { return Contra{b.params->contra}; }
  static __device__ __forceinline__ float twoloop(const TreeBatch& b, uint32_t a, uint32_t bb, int ci,
                                                  int cj, int x1, int /*x2*/, int y1, int /*y2*/,
                                                  int ak, int al, int m2, int m3) {
    return contra_twoloop_flat(b.params->contra, a, bb, ci, cj, x1, y1, ak, al, m2, m3);
  }
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
    // 2-bit packed copy of the sequence: base p at bit 2(p+32); zeros around it
    uint32_t* pk = reinterpret_cast<uint32_t*>(b.workspace + sd.pk_off);
    const uint8_t* s = b.bases + sd.seq_off;
    for (size_t wd = t0; wd < sd.pk_words; wd += stride) {
      uint32_t v = 0;
      for (uint32_t y = 0; y < 16; y++) {
        const int64_t pos = static_cast<int64_t>(wd) * 16 + y - 32;
        if (pos >= 0 && pos < static_cast<int64_t>(sd.n)) v |= static_cast<uint32_t>(s[pos] & 3u) << (2u * y);
      }
      pk[wd] = v;
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
// 2-loop terms.  The <= 496 probes of a cell sit in 512 slots spread over the cell's lanes.  A
// generic loop costs a table lookup: length part of the slot + the fixed pair's side (uniform
// over the cell, by class) + the varying pair's side from a 256-entry table indexed by the four
// bases around that pair, which are adjacent 2-bit fields of the two base windows.  The few
// small loops with explicit tables (Turner: stack, 0x1, 1x1, 1x2, 2x1, 2x2; CONTRAfold: stack,
// 0x1, 1x1) are scored by the flat scorers of rnamc_scoring.h in one extra pass of the cell's
// first lanes.
template <bool CONTRA>
struct Special;
template <>
struct Special<false> {
  static constexpr uint32_t N = 7;
  static __device__ __forceinline__ void slot(uint32_t t, uint32_t& a, uint32_t& bb) {
    // (0,0) (0,1) (1,0) (1,1) (1,2) (2,1) (2,2)
    a = t < 2u ? 0u : (t < 5u ? 1u : 2u);
    bb = t == 0u ? 0u : (t == 1u ? 1u : (t == 2u ? 0u : (t == 3u ? 1u : (t == 4u ? 2u : (t == 5u ? 1u : 2u)))));
  }
};
template <>
struct Special<true> {
  static constexpr uint32_t N = 4;
  static __device__ __forceinline__ void slot(uint32_t t, uint32_t& a, uint32_t& bb) {
    // (0,0) (0,1) (1,0) (1,1)
    a = t >> 1;
    bb = t & 1u;
  }
};

__device__ __forceinline__ float sel4(uint32_t c, float v0, float v1, float v2, float v3) {
  return c == 0u ? v0 : (c == 1u ? v1 : (c == 2u ? v2 : v3));
}

// side of a FIXED pair (p0,p1) with the two bases q0 next to p0 and q1 next to p1 on the loop
// side, per class: Turner pen | mismatch_c[p0][p1][q0][q1] + pen; CONTRAfold helix_close +
// terminal_mismatch (+ extra: the base-pair score when the pair is the enclosed one)
template <bool CONTRA>
__device__ __forceinline__ void fixed_side(const TreeBatch& b, int p0, int p1, int q0, int q1, float extra,
                                           float (&v)[4]) {
  if (CONTRA) {
    const rnamc_fold_score_sets& f = b.params->contra;
    const float x = (f.helix_close_scores[p0][p1] + f.terminal_mismatch_scores[p0][p1][q0][q1]) + extra;
    v[0] = v[1] = v[2] = v[3] = x;
  } else {
    const rnamc_turner_scores& tt = b.params->turner;
    const float pen = augu(p0, p1) ? tt.helix_augu_end_penalty : 0.f;
    v[0] = pen;
    v[1] = tt.terminal_mismatch_scores_1xmany[p0][p1][q0][q1] + pen;
    v[2] = tt.terminal_mismatch_scores_2x3[p0][p1][q0][q1] + pen;
    v[3] = tt.terminal_mismatch_scores_interior[p0][p1][q0][q1] + pen;
  }
}

// closing-pair block of cell (i,j): hairpin and multibranch term (both held by the caller,
// uniform) and the <= 496 enclosed pairs (src/mccaskill_algo.rs:306-325 / 412-436);
// wi = bases i .. i+31, wj = bases j-31 .. j
template <bool CONTRA, int TPC>
__device__ __forceinline__ void pair_block(const TreeBatch& b, const float* __restrict__ qb_r,
                                           uint32_t ld, Acc& acc, uint32_t i, uint32_t j, uint32_t t,
                                           uint64_t wi, uint64_t wj, float hp, float mbt) {
  const uint32_t d = j - i;
  acc_add(acc, t == 0u ? hp : (t == 1u ? mbt : kNegInf));
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 1) return;
#endif
  if (d < 3u) return;
  const int ci = wb(wi, 0), cj = wb(wj, 31);
  const int x1 = wb(wi, 1), x2 = wb(wi, 2), y1 = wb(wj, 30), y2 = wb(wj, 29);
  float cs[4];
  fixed_side<CONTRA>(b, ci, cj, x1, y1, 0.f, cs);
  const float* __restrict__ tin = &b.tabs->in[CONTRA ? 1 : 0][0][0];
  const float* __restrict__ tlen = &b.tabs->len[CONTRA ? 1 : 0][0];
  const uint32_t* __restrict__ tslot = &b.tabs->slot[CONTRA ? 1 : 0][0];
#pragma unroll
  for (uint32_t p = t; p < 512u; p += TPC) {
    const uint32_t sl = tslot[p];
    const float ln = tlen[p];
    const uint32_t a = sl & 255u, bb = (sl >> 8) & 255u, cls = (sl >> 16) & 255u;
    if ((sl >> 24) == 1u && a + bb + 3u <= d) {
      const uint32_t k = i + 1u + a, l = j - 1u - bb;
      const float x = qb_r[static_cast<size_t>(k) * ld + l];
      const uint32_t idx = (static_cast<uint32_t>(wi >> (2u * a)) & 15u) |
                           ((static_cast<uint32_t>(wj >> (60u - 2u * bb)) & 15u) << 4);
      const float sc = (ln + sel4(cls, cs[0], cs[1], cs[2], cs[3])) + tin[cls * 256u + idx];
      acc_add(acc, x + sc);  // (absent pair: x = -inf)
    }
  }
  if (t < Special<CONTRA>::N) {
    uint32_t a, bb;
    Special<CONTRA>::slot(t, a, bb);
    if (a + bb + 3u <= d) {
      const uint32_t k = i + 1u + a, l = j - 1u - bb;
      const float x = qb_r[static_cast<size_t>(k) * ld + l];
      // (k,l) = bases at window positions 1+a / 30-bb; their outer neighbours a / 31-bb
      const float sc = TModel<CONTRA>::twoloop(b, a, bb, ci, cj, x1, x2, y1, y2, wb(wi, 1u + a),
                                               wb(wj, 30u - bb), wb(wj, 31u - bb), wb(wi, a));
      acc_add(acc, x + sc);
    }
  }
}

// enclosing 2-loops of the finished pair (i,j) (574-593 / 681-700): (k,l) = (i-1-a, j+1+b)
// closes, (i,j) is enclosed; operands {log bpp, sums_close}(k,l) in one 8-byte load
template <bool CONTRA, int TPC>
__device__ __forceinline__ void outer_block(const TreeBatch& b, const TSeq& q, Acc& acc, uint32_t i,
                                            uint32_t j, uint32_t t, float qb) {
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 1) return;
#endif
  const uint32_t n = q.n, ld = q.ld;
  const float2* __restrict__ pq_r = reinterpret_cast<const float2*>(q.m[T_PQ]);
  // bases i-31 .. i and j .. j+31
  const uint64_t wi = load_win64(q.pk, static_cast<int>(i) - 31);
  const uint64_t wj = load_win64(q.pk, static_cast<int>(j));
  const int ai = wb(wi, 31), aj = wb(wj, 0), m3 = wb(wi, 30), m2 = wb(wj, 1);
  float es[4];
  fixed_side<CONTRA>(b, aj, ai, m2, m3, CONTRA ? b.params->contra.basepair_scores[ai][aj] : 0.f, es);
  const float* __restrict__ tout = &b.tabs->out[CONTRA ? 1 : 0][0][0];
  const float* __restrict__ tlen = &b.tabs->len[CONTRA ? 1 : 0][0];
  const uint32_t* __restrict__ tslot = &b.tabs->slot[CONTRA ? 1 : 0][0];
#pragma unroll
  for (uint32_t p = t; p < 512u; p += TPC) {
    const uint32_t sl = tslot[p];
    const float ln = tlen[p];
    const uint32_t a = sl & 255u, bb = (sl >> 8) & 255u, cls = (sl >> 16) & 255u;
    if ((sl >> 24) == 1u && a < i && j + 1u + bb < n) {
      const uint32_t k = i - 1u - a, l = j + 1u + bb;
      const float2 pq = pq_r[static_cast<size_t>(k) * ld + l];
      const uint32_t idx = (static_cast<uint32_t>(wi >> (60u - 2u * a)) & 15u) |
                           ((static_cast<uint32_t>(wj >> (2u * bb)) & 15u) << 4);
      const float sc = (ln + sel4(cls, es[0], es[1], es[2], es[3])) + tout[cls * 256u + idx];
      if (pq.y > kNegInf) acc_add(acc, ((pq.x + qb) - pq.y) + sc);
    }
  }
  if (t < Special<CONTRA>::N) {
    uint32_t a, bb;
    Special<CONTRA>::slot(t, a, bb);
    if (a < i && j + 1u + bb < n) {
      const uint32_t k = i - 1u - a, l = j + 1u + bb;
      const float2 pq = pq_r[static_cast<size_t>(k) * ld + l];
      // closing pair at window positions 30-a / 1+bb; its inner neighbours 31-a, 32-a / bb, bb-1
      const float sc = TModel<CONTRA>::twoloop(b, a, bb, wb(wi, 30u - a), wb(wj, 1u + bb), wb(wi, 31u - a),
                                               wb(wi, a >= 1u ? 32u - a : 31u), wb(wj, bb),
                                               wb(wj, bb >= 1u ? bb - 1u : 0u), ai, aj, m2, m3);
      if (pq.y > kNegInf) acc_add(acc, ((pq.x + qb) - pq.y) + sc);
    }
  }
}

// ----------------------------------------------------------------------------
// inside pass.  TPC threads share the cell (i, i+d): 64 (four cells per workgroup, no
// barrier), 256 or 1024 (one cell per workgroup); the host picks by how many cells the
// diagonal holds.  Scalars of the cell are computed by all of its threads alike (uniform
// loads issued at the top, beside the operand streams); its first lane stores.
template <bool CONTRA, int TPC>
__global__ void __launch_bounds__(TPC < 256 ? 256 : TPC) k_tree_inside(TreeBatch b, uint32_t d) {
  constexpr int BLOCK = TPC < 256 ? 256 : TPC;
  __shared__ float red[BLOCK / 64][4][2];
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 4) return;
#endif
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  // (wave-uniform: TPC is a multiple of 64; readfirstlane lets the cell's scalar work run on
  // the scalar unit)
  const uint32_t i = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
      static_cast<int>(blockIdx.x * (BLOCK / TPC) + threadIdx.x / TPC)));
  if (i + d >= n) return;
  const uint32_t j = i + d;
  const uint32_t t = threadIdx.x % TPC;
  const uint8_t* __restrict__ s = q.s;
  const auto model = TModel<CONTRA>::make(b);
  const float* __restrict__ qb_r = q.m[T_QB];
  const size_t row_i = static_cast<size_t>(i) * ld, col_j = static_cast<size_t>(j) * ld;
  // bases i .. i+31 and j-31 .. j
  const uint64_t wi = load_win64(q.pk, static_cast<int>(i));
  const uint64_t wj = load_win64(q.pk, static_cast<int>(j) - 31);
  const int ci = wb(wi, 0), cj = wb(wj, 31);

  bool act = canonical(ci, cj);
  if (!(b.allows_short_hairpins && CONTRA) && d + 1 < RNAMC_MIN_SPAN_HAIRPIN_CLOSE) act = false;

  // operands of the cell's scalar recurrences (uniform loads)
  float zr_e_prev = kNegInf, zr_m_prev = kNegInf;
  if (j >= 1) {
    zr_e_prev = q.m[T_ZRE][col_j - ld + i];
    if (CONTRA) zr_m_prev = q.m[T_ZRM][col_j - ld + i];
  }
  const float u_next = q.m[T_U][col_j + i + 1];  // (i+1 == n: the column's pad, -inf)
  const float zs_next = (j == n - 1) ? q.zs[i + 1] : 0.f;
  float hp = kNegInf, mbt = kNegInf, accs = 0.f;
  if (act) {
    if (!CONTRA || d - 1 <= RNAMC_MAX_LOOP_LEN) hp = model.hairpin(s, n, i, j);
    if (d >= 2) mbt = q.m[T_QM][row_i + ld + (j - 1)] + model.mbclose(s, n, i, j);
    accs = model.accessible(s, n, i, j);
  }

  Acc acc[4] = {acc_empty(), acc_empty(), acc_empty(), acc_empty()};
  // [0] closing-pair block (297-343 / 400-467)
  if (act) pair_block<CONTRA, TPC>(b, qb_r, ld, acc[0], i, j, t, wi, wj, hp, mbt);
  // [1] sums_multibranch (L_c second fold): k = i+1 .. j-1, Q1(i,k-1) + Zr_mb(k,j)
#ifdef RNAMC_DEBUG_KNOBS
  if (!(b.debug & 2))
#endif
  if (d >= 2) acc_product<TPC>(acc[1], q.m[T_Q1R] + row_i + i, q.m[T_ZRM] + col_j + i + 1, d - 1, t);
  // [2] Z(0,j): k = 1 .. j, Zr_ext(k,j) + Z(0,k-1)       (the k = 0 term is this cell's own)
  if (i == 0 && j >= 1) acc_product<TPC>(acc[2], q.m[T_ZRE] + col_j + 1, q.zp + 1, j, t);
  // [3] Z(i,n-1): l = i+1 .. n-2, Qa(i,l) + Z(l+1,n-1)   (l = n-1 is this cell's own)
  if (j == n - 1 && d >= 1) acc_product<TPC>(acc[3], q.m[T_QA] + row_i + i + 1, q.zs + i + 2, d - 1, t);
  if (!cell_reduce<4, TPC>(acc, red)) return;

  const float ext_bp = CONTRA ? b.params->contra.external_score_basepair : 0.f;
  const float ext_un = CONTRA ? b.params->contra.external_score_unpair : 0.f;
  const float mb_bp = CONTRA ? b.params->contra.multibranch_score_basepair
                             : b.params->turner.coeff_num_branches;
  const float mb_un = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  const bool st = t == 0u;  // the lane that stores
  float qa = kNegInf;
  if (act) {
    const float qb = acc_value(acc[0]);
    if (qb > kNegInf) {
      qa = qb + accs;
      if (st) {
        q.m[T_QB][row_i + j] = qb;
        q.m[T_QA][row_i + j] = qa;
      }
    }
  }
  // sums_rightmost_basepairs_{external,multibranch}: one step from the cell to the left
  const float zr_e = lse2(zr_e_prev + ext_un, qa + ext_bp);
  const float zr_m = CONTRA ? lse2(zr_m_prev + mb_un, qa + mb_bp) : zr_e + mb_bp;
  const float u = lse2(u_next + mb_un, zr_m);
  const float qm = acc_value(acc[1]);
  const float q1 = lse2(u, qm);
  if (st) {
    q.m[T_ZRE][col_j + i] = zr_e;
    q.m[T_ZRM][col_j + i] = zr_m;
    q.m[T_U][col_j + i] = u;
    q.m[T_QM][row_i + j] = qm;
    q.m[T_Q1R][row_i + j] = q1;
    q.m[T_Q1C][col_j + i] = q1;
  }
  if (i == 0) {
    // sums_external[0][j] (352-363 / 487-498)
    Acc z = acc[2];
    acc_add(z, zr_e);  // k = 0: Z(0,-1) = 0
    acc_add(z, CONTRA ? ext_un * static_cast<float>(j + 1) : 0.f);
    if (st) q.zp[j + 1] = acc_value(z);
  }
  if (j == n - 1) {
    Acc z = acc[3];
    z.m += ext_bp;            // every product term carries the pair's external_score_basepair
    acc_add(z, qa + ext_bp);  // l = n-1: Z(n,n-1) = 0
    acc_add(z, zs_next + ext_un);
    if (st) q.zs[i] = acc_value(z);
  }
}

// ----------------------------------------------------------------------------
// Two diagonals per launch.  The sweep's cost is its NUMBER of dependent launches (each a
// few microseconds of launch gap, operand round trips and drain), so one workgroup takes the
// cells (i, j) and (i, j+1) of diagonals d and d+1.  What diagonal d+1 needs of diagonal d:
//   inside : Zr(i,j) (own cell) and U(i+1,j+1), one step from U(i+2,j+1) given the
//            closing-pair block of the neighbour (i+1,j+1), which is evaluated a second time here
//            (sums_multibranch(i,j+1) itself does not: its k = i+1 term carries Q1(i,i) = -inf);
//   outside: W(i,j+1), Pm2(i,j+1) (own cell) and probs_multibranch(i-1,j) for the column
//            prefix, a product that shares its Q1 row with the cell's own and is evaluated here
//            as a third stream (the k = j+1 term of Pm(i,j) carries Q1(j+1,j) = -inf, and the
//            k = i-1 term of L_e Q1(i,i-1) = -inf: neither needs the neighbour).

// (+)_k over idx in [0, len1): a = A[idx]; acc0 += a + B0[idx] (idx < len0); acc1 += a + B1[idx]
template <int TPC>
__device__ __forceinline__ void acc_product_2b(Acc& acc0, Acc& acc1, const float* __restrict__ A,
                                               const float* __restrict__ B0,
                                               const float* __restrict__ B1, uint32_t len0,
                                               uint32_t len1, uint32_t t) {
  for (uint32_t k = t; k < len1; k += 2u * TPC) {
    const uint32_t k1 = k + TPC;
    const bool v1 = k1 < len1;
    const float a0 = A[k], a1 = v1 ? A[k1] : kNegInf;
    const float p0 = k < len0 ? B0[k] : kNegInf, p1 = k1 < len0 ? B0[k1] : kNegInf;
    const float r0 = B1[k], r1 = v1 ? B1[k1] : kNegInf;
    acc_add2(acc0, a0 + p0, a1 + p1);
    acc_add2(acc1, a0 + r0, a1 + r1);
  }
}

template <bool CONTRA, int TPC>
__global__ void __launch_bounds__(TPC < 256 ? 256 : TPC) k_tree_inside2(TreeBatch b, uint32_t d) {
  constexpr int BLOCK = TPC < 256 ? 256 : TPC;
  constexpr int NA = 9;
  __shared__ float red[BLOCK / 64][NA][2];
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 4) return;
#endif
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  const uint32_t i = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
      static_cast<int>(blockIdx.x * (BLOCK / TPC) + threadIdx.x / TPC)));
  if (i + d >= n) return;
  const uint32_t j = i + d, j1 = j + 1u;
  const bool has1 = j1 < n;  // cells (i, j+1) and (i+1, j+1) exist
  const uint32_t t = threadIdx.x % TPC;
  const uint8_t* __restrict__ s = q.s;
  const auto model = TModel<CONTRA>::make(b);
  const float* __restrict__ qb_r = q.m[T_QB];
  const size_t row_i = static_cast<size_t>(i) * ld, col_j = static_cast<size_t>(j) * ld;
  const uint64_t wi = load_win64(q.pk, static_cast<int>(i));          // bases i .. i+31
  const uint64_t wi1 = load_win64(q.pk, static_cast<int>(i) + 1);     // bases i+1 .. i+32
  const uint64_t wj = load_win64(q.pk, static_cast<int>(j) - 31);     // bases j-31 .. j
  const uint64_t wj1 = load_win64(q.pk, static_cast<int>(j) - 30);    // bases j-30 .. j+1
  const bool span_ok = (b.allows_short_hairpins && CONTRA);
  const bool act0 = canonical(wb(wi, 0), wb(wj, 31)) && (span_ok || d + 1 >= RNAMC_MIN_SPAN_HAIRPIN_CLOSE);
  const bool act1 = has1 && canonical(wb(wi, 0), wb(wj1, 31)) && (span_ok || d + 2 >= RNAMC_MIN_SPAN_HAIRPIN_CLOSE);
  const bool actn = has1 && canonical(wb(wi1, 0), wb(wj1, 31)) && (span_ok || d + 1 >= RNAMC_MIN_SPAN_HAIRPIN_CLOSE);

  // operands of the scalar recurrences (uniform loads, all from diagonals < d)
  float zr_e_prev = kNegInf, zr_m_prev = kNegInf, zr_e_prevn = kNegInf, zr_m_prevn = kNegInf;
  if (j >= 1) {
    zr_e_prev = q.m[T_ZRE][col_j - ld + i];
    if (CONTRA) zr_m_prev = q.m[T_ZRM][col_j - ld + i];
  }
  const float u_next0 = q.m[T_U][col_j + i + 1];            // U(i+1, j)
  float u_nextn = kNegInf;                                   // U(i+2, j+1)
  if (has1) {
    zr_e_prevn = q.m[T_ZRE][col_j + i + 1];                  // Zr_ext(i+1, j)
    if (CONTRA) zr_m_prevn = q.m[T_ZRM][col_j + i + 1];
    u_nextn = q.m[T_U][col_j + ld + i + 2];
  }
  float hp0 = kNegInf, mbt0 = kNegInf, accs0 = 0.f, hp1 = kNegInf, mbt1 = kNegInf, accs1 = 0.f,
        hpn = kNegInf, mbtn = kNegInf, accsn = 0.f;
  if (act0) {
    if (!CONTRA || d - 1 <= RNAMC_MAX_LOOP_LEN) hp0 = model.hairpin(s, n, i, j);
    if (d >= 2) mbt0 = q.m[T_QM][row_i + ld + (j - 1)] + model.mbclose(s, n, i, j);
    accs0 = model.accessible(s, n, i, j);
  }
  if (act1) {
    if (!CONTRA || d <= RNAMC_MAX_LOOP_LEN) hp1 = model.hairpin(s, n, i, j1);
    if (d >= 1) mbt1 = q.m[T_QM][row_i + ld + j] + model.mbclose(s, n, i, j1);
    accs1 = model.accessible(s, n, i, j1);
  }
  if (actn) {
    if (!CONTRA || d - 1 <= RNAMC_MAX_LOOP_LEN) hpn = model.hairpin(s, n, i + 1, j1);
    if (d >= 2) mbtn = q.m[T_QM][row_i + 2 * static_cast<size_t>(ld) + j] + model.mbclose(s, n, i + 1, j1);
    accsn = model.accessible(s, n, i + 1, j1);
  }
  const bool zs0 = j == n - 1, zs1 = has1 && j1 == n - 1;  // which cell sits in column n-1
  const float zs_a = zs0 ? q.zs[i + 1] : (zs1 ? q.zs[i + 2] : 0.f);  // Z(i+1,n-1) | Z(i+2,n-1)

  Acc acc[NA];
#pragma unroll
  for (int x = 0; x < NA; x++) acc[x] = acc_empty();
  // [0] [1] [2] closing-pair blocks of (i,j), (i+1,j+1), (i,j+1)
  if (act0) pair_block<CONTRA, TPC>(b, qb_r, ld, acc[0], i, j, t, wi, wj, hp0, mbt0);
  if (actn) pair_block<CONTRA, TPC>(b, qb_r, ld, acc[1], i + 1, j1, t, wi1, wj1, hpn, mbtn);
  if (act1) pair_block<CONTRA, TPC>(b, qb_r, ld, acc[2], i, j1, t, wi, wj1, hp1, mbt1);
  // [3] [4] sums_multibranch of (i,j) and (i,j+1): k = i+1 .. j-1 | j, Q1(i,k-1) + Zr_mb(k,j | j+1)
  // (the k = i+1 term of the second reads a cell of this launch: masked, it carries Q1(i,i) = -inf)
#ifdef RNAMC_DEBUG_KNOBS
  if (!(b.debug & 2))
#endif
  {
    if (has1) {
      if (d >= 2)
        acc_product_2b<TPC>(acc[3], acc[4], q.m[T_Q1R] + row_i + i + 1, q.m[T_ZRM] + col_j + i + 2,
                            q.m[T_ZRM] + col_j + ld + i + 2, d - 2, d - 1, t);
      // (k = i+1 of the first, left out above to keep both streams on the same k: one lane)
      if (d >= 2 && t == 0u) acc_add(acc[3], q.m[T_Q1R][row_i + i] + q.m[T_ZRM][col_j + i + 1]);
    } else if (d >= 2) {
      acc_product<TPC>(acc[3], q.m[T_Q1R] + row_i + i, q.m[T_ZRM] + col_j + i + 1, d - 1, t);
    }
    // [5] [6] Z(0,j), Z(0,j+1): k >= 1 | 2, Zr_ext(k,.) + Z(0,k-1)
    if (i == 0) {
      if (j >= 1) acc_product<TPC>(acc[5], q.m[T_ZRE] + col_j + 1, q.zp + 1, j, t);
      if (has1 && j1 >= 2) acc_product<TPC>(acc[6], q.m[T_ZRE] + col_j + ld + 2, q.zp + 2, j1 - 1, t);
    }
    // [7] column n-1 cell of this workgroup: l = i+1 .. n-2, Qa(i,l) + Z(l+1,n-1)
    //     (zs0: (i,j), all l from memory; zs1: (i,j+1), l = j = n-2 is this workgroup's own cell)
    if (zs0 && d >= 1) acc_product<TPC>(acc[7], q.m[T_QA] + row_i + i + 1, q.zs + i + 2, d - 1, t);
    if (zs1 && d >= 1) acc_product<TPC>(acc[7], q.m[T_QA] + row_i + i + 1, q.zs + i + 2, d - 1, t);
    // [8] zs1: the neighbour (i+1,n-1)'s own sum, l = i+2 .. n-2
    if (zs1 && d >= 1)
      acc_product<TPC>(acc[8], q.m[T_QA] + row_i + ld + i + 2, q.zs + i + 3, d - 1, t);
  }
  if (!cell_reduce<NA, TPC>(acc, red)) return;

  const float ext_bp = CONTRA ? b.params->contra.external_score_basepair : 0.f;
  const float ext_un = CONTRA ? b.params->contra.external_score_unpair : 0.f;
  const float mb_bp = CONTRA ? b.params->contra.multibranch_score_basepair
                             : b.params->turner.coeff_num_branches;
  const float mb_un = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  const bool st = t == 0u;  // the lane that stores
  // ---- cell (i,j)
  float qa0 = kNegInf;
  if (act0) {
    const float qb = acc_value(acc[0]);
    if (qb > kNegInf) {
      qa0 = qb + accs0;
      if (st) {
        q.m[T_QB][row_i + j] = qb;
        q.m[T_QA][row_i + j] = qa0;
      }
    }
  }
  const float zr_e0 = lse2(zr_e_prev + ext_un, qa0 + ext_bp);
  const float zr_m0 = CONTRA ? lse2(zr_m_prev + mb_un, qa0 + mb_bp) : zr_e0 + mb_bp;
  const float u0 = lse2(u_next0 + mb_un, zr_m0);
  const float qm0 = acc_value(acc[3]);
  const float q1_0 = lse2(u0, qm0);
  if (st) {
    q.m[T_ZRE][col_j + i] = zr_e0;
    q.m[T_ZRM][col_j + i] = zr_m0;
    q.m[T_U][col_j + i] = u0;
    q.m[T_QM][row_i + j] = qm0;
    q.m[T_Q1R][row_i + j] = q1_0;
    q.m[T_Q1C][col_j + i] = q1_0;
  }
  float zp_j = 0.f;  // Z(0,j)
  if (i == 0) {
    Acc z = acc[5];
    acc_add(z, zr_e0);  // k = 0: Z(0,-1) = 0
    acc_add(z, CONTRA ? ext_un * static_cast<float>(j + 1) : 0.f);
    zp_j = acc_value(z);
    if (st) q.zp[j + 1] = zp_j;
  }
  if (zs0) {
    Acc z = acc[7];
    z.m += ext_bp;             // every product term carries the pair's external_score_basepair
    acc_add(z, qa0 + ext_bp);  // l = n-1: Z(n,n-1) = 0
    acc_add(z, zs_a + ext_un);
    if (st) q.zs[i] = acc_value(z);
  }
  if (!has1) return;
  // ---- neighbour (i+1,j+1): closing pair -> Zr -> U, not stored (its own workgroup does)
  float qan = kNegInf;
  if (actn) {
    const float qb = acc_value(acc[1]);
    if (qb > kNegInf) qan = qb + accsn;
  }
  const float zr_en = lse2(zr_e_prevn + ext_un, qan + ext_bp);
  const float zr_mn = CONTRA ? lse2(zr_m_prevn + mb_un, qan + mb_bp) : zr_en + mb_bp;
  const float un = lse2(u_nextn + mb_un, zr_mn);  // U(i+1, j+1)
  // ---- cell (i,j+1)
  const size_t col_j1 = col_j + ld;
  float qa1 = kNegInf;
  if (act1) {
    const float qb = acc_value(acc[2]);
    if (qb > kNegInf) {
      qa1 = qb + accs1;
      if (st) {
        q.m[T_QB][row_i + j1] = qb;
        q.m[T_QA][row_i + j1] = qa1;
      }
    }
  }
  const float zr_e1 = lse2(zr_e0 + ext_un, qa1 + ext_bp);
  const float zr_m1 = CONTRA ? lse2(zr_m0 + mb_un, qa1 + mb_bp) : zr_e1 + mb_bp;
  const float u1 = lse2(un + mb_un, zr_m1);
  const float qm1 = acc_value(acc[4]);
  const float q1_1 = lse2(u1, qm1);
  if (st) {
    q.m[T_ZRE][col_j1 + i] = zr_e1;
    q.m[T_ZRM][col_j1 + i] = zr_m1;
    q.m[T_U][col_j1 + i] = u1;
    q.m[T_QM][row_i + j1] = qm1;
    q.m[T_Q1R][row_i + j1] = q1_1;
    q.m[T_Q1C][col_j1 + i] = q1_1;
  }
  if (i == 0) {
    Acc z = acc[6];
    acc_add(z, zr_e1);                // k = 0
    acc_add(z, zr_en + q.zp[1]);      // k = 1: Zr_ext(1,j+1) + Z(0,0)
    acc_add(z, CONTRA ? ext_un * static_cast<float>(j1 + 1) : 0.f);
    if (st) q.zp[j1 + 1] = acc_value(z);
  }
  if (zs1) {
    // Z(i+1,n-1) of the neighbour first: its own cell term, its memory terms, Z(i+2,n-1)
    Acc zn = acc[8];
    zn.m += ext_bp;
    acc_add(zn, qan + ext_bp);
    acc_add(zn, zs_a + ext_un);
    const float zs_n = acc_value(zn);
    // acc[7] holds the memory terms l = i+1 .. j-1; l = j = n-2 is this launch's own cell (i,j)
    Acc z = acc[7];
    z.m += ext_bp;
    acc_add(z, (qa0 + ext_bp) + q.zs[j + 1]);  // l = j = n-2: Z(n-1,n-1)
    acc_add(z, qa1 + ext_bp);                  // l = n-1
    acc_add(z, zs_n + ext_un);
    if (st) q.zs[i] = acc_value(z);
  }
}

// ----------------------------------------------------------------------------
// outside pass, same thread layout
template <bool CONTRA, int TPC>
__global__ void __launch_bounds__(TPC < 256 ? 256 : TPC) k_tree_outside(TreeBatch b, uint32_t d) {
  constexpr int BLOCK = TPC < 256 ? 256 : TPC;
  __shared__ float red[BLOCK / 64][3][2];
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 4) return;
#endif
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  const uint32_t i = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
      static_cast<int>(blockIdx.x * (BLOCK / TPC) + threadIdx.x / TPC)));
  if (i + d >= n) return;
  const uint32_t j = i + d;
  const uint32_t t = threadIdx.x % TPC;
  const uint8_t* __restrict__ s = q.s;
  const auto model = TModel<CONTRA>::make(b);
  const float* __restrict__ qb_r = q.m[T_QB];
  const size_t row_i = static_cast<size_t>(i) * ld, col_j = static_cast<size_t>(j) * ld;
  const float* __restrict__ w_r = q.m[T_ZRE];   // W = (P + mbclose) - Qb, row-major
  float* __restrict__ r_c = q.m[T_ZRM];         // R = Pm (+) Pm2, column-major
  float* __restrict__ pm2_r = q.m[T_QM];        // probs_multibranch2, row-major
  float* __restrict__ sp_c = q.m[T_U];          // sp_c(i,j) = (+)_{k<=i} Pm(k,j) [+ unpaired], column-major

  const float qb = qb_r[row_i + j];
  const bool paired = qb > kNegInf;  // (uniform over the cell's threads)
  float pm2_next = kNegInf, w_next = kNegInf, sp_prev = kNegInf;
  if (j + 1 < n) {
    pm2_next = pm2_r[row_i + j + 1];
    w_next = w_r[row_i + j + 1];
  }
  if (i >= 1) sp_prev = sp_c[col_j + i - 1];
  float qa = kNegInf, mbc = 0.f, zpi = 0.f, zsj = 0.f, ztot = 0.f;
  if (paired) {
    qa = q.m[T_QA][row_i + j];
    mbc = model.mbclose(s, n, i, j);
    zpi = q.zp[i];
    zsj = q.zs[j + 1];
    ztot = q.zp[n];
  }
  Acc acc[3] = {acc_empty(), acc_empty(), acc_empty()};
  // [0] probs_multibranch (L_d): k = j+1 .. n-1, W(i,k) + Q1(j+1,k-1)
#ifdef RNAMC_DEBUG_KNOBS
  if (!(b.debug & 2))
#endif
  if (j + 1 < n)
    acc_product<TPC>(acc[0], w_r + row_i + j + 1, q.m[T_Q1R] + static_cast<size_t>(j + 1) * ld + j,
                     n - 1 - j, t);
  if (paired) {
    // [1] enclosing 2-loops
    outer_block<CONTRA, TPC>(b, q, acc[1], i, j, t, qb);
    // [2] L_e cases one and three: k = 0 .. i-1, Q1(k+1,i-1) + R(k,j)
#ifdef RNAMC_DEBUG_KNOBS
    if (!(b.debug & 2))
#endif
    if (i >= 1) acc_product<TPC>(acc[2], q.m[T_Q1C] + static_cast<size_t>(i - 1) * ld + 1, r_c + col_j, i, t);
  }
  if (!cell_reduce<3, TPC>(acc, red)) return;

  const bool st = t == 0u;
  const float mb_un = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  const float pm = acc_value(acc[0]);
  const float pm2 = lse2(pm2_next + mb_un, w_next);
  if (st) {
    pm2_r[row_i + j] = pm2;
    r_c[col_j + i] = lse2(pm, pm2);
    sp_c[col_j + i] = lse2(sp_prev + mb_un, pm);
  }
  if (!paired) return;
  // exterior term (561-573 / 676-680)
  const float ext = CONTRA ? (((zpi + zsj) + qa) + b.params->contra.external_score_basepair) - ztot
                           : ((zpi + qa) + zsj) - ztot;
  Acc pa = acc[1];
  acc_add(pa, ext);
  // L_e: every term carries A = Qa + (coeff_num_branches | multibranch_score_basepair)
  const float A = qa + (CONTRA ? b.params->contra.multibranch_score_basepair
                               : b.params->turner.coeff_num_branches);
  acc_add(pa, A + acc_value(acc[2]));
  acc_add(pa, A + sp_prev);  // case two: sp_prev holds the unpaired factors of rows k < i already
  const float lp = acc_value(pa);
  if (st && lp > kNegInf) {
    q.out[tri_off(n, d) + i] = lp;
    q.m[T_ZRE][row_i + j] = (lp + mbc) - qb;
    reinterpret_cast<float2*>(q.m[T_PQ])[row_i + j] = make_float2(lp, qb);
  }
}

// (+) over idx in [0, len0): three products off four streams
//   pm0 += Wi[idx] + Qa[idx]          (cell (i,j):     k = j+1+idx)
//   pmn += Wm[idx] + Qa[idx]          (cell (i-1,j),   do_n)
//   pm1 += Wi[idx] + Qb[idx]          (cell (i,j+1):   idx >= 1, do_1)
template <int TPC>
__device__ __forceinline__ void acc_product_3(Acc& pm0, Acc& pmn, Acc& pm1,
                                              const float* __restrict__ Wi,
                                              const float* __restrict__ Wm,
                                              const float* __restrict__ Qa,
                                              const float* __restrict__ Qb, uint32_t len0, bool do_n,
                                              bool do_1, uint32_t t) {
  for (uint32_t k = t; k < len0; k += 2u * TPC) {
    const uint32_t k1 = k + TPC;
    const bool v1 = k1 < len0;
    const float wi0 = Wi[k], wi1 = v1 ? Wi[k1] : kNegInf;
    const float qa0 = Qa[k], qa1 = v1 ? Qa[k1] : kNegInf;
    acc_add2(pm0, wi0 + qa0, wi1 + qa1);
    if (do_n) {
      const float wm0 = Wm[k], wm1 = v1 ? Wm[k1] : kNegInf;
      acc_add2(pmn, wm0 + qa0, wm1 + qa1);
    }
    if (do_1) {
      const float qb0 = k >= 1u ? Qb[k] : kNegInf, qb1 = v1 ? Qb[k1] : kNegInf;
      acc_add2(pm1, wi0 + qb0, wi1 + qb1);
    }
  }
}

// diagonals d+1 (cell (i,j+1)) and d (cell (i,j)) of the outside sweep
template <bool CONTRA, int TPC>
__global__ void __launch_bounds__(TPC < 256 ? 256 : TPC) k_tree_outside2(TreeBatch b, uint32_t d) {
  constexpr int BLOCK = TPC < 256 ? 256 : TPC;
  constexpr int NA = 7;
  __shared__ float red[BLOCK / 64][NA][2];
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 4) return;
#endif
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  const uint32_t i = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
      static_cast<int>(blockIdx.x * (BLOCK / TPC) + threadIdx.x / TPC)));
  if (i + d >= n) return;
  const uint32_t j = i + d, j1 = j + 1u;
  const bool has1 = j1 < n;
  const uint32_t t = threadIdx.x % TPC;
  const uint8_t* __restrict__ s = q.s;
  const auto model = TModel<CONTRA>::make(b);
  const float* __restrict__ qb_r = q.m[T_QB];
  const size_t row_i = static_cast<size_t>(i) * ld, col_j = static_cast<size_t>(j) * ld;
  const float* __restrict__ w_r = q.m[T_ZRE];   // W = (P + mbclose) - Qb, row-major
  float* __restrict__ r_c = q.m[T_ZRM];         // R = Pm (+) Pm2, column-major
  float* __restrict__ pm2_r = q.m[T_QM];        // probs_multibranch2, row-major
  float* __restrict__ sp_c = q.m[T_U];          // sp_c(i,j) = (+)_{k<=i} Pm(k,j) [+ unpaired], column-major

  const float qb0 = qb_r[row_i + j];
  const float qb1 = has1 ? qb_r[row_i + j1] : kNegInf;
  const bool paired0 = qb0 > kNegInf, paired1 = qb1 > kNegInf;  // (uniform)
  // operands of the scalar recurrences, all from diagonals >= d+2
  float pm2_next1 = kNegInf, w_next1 = kNegInf, sp_prev1 = kNegInf, sp_prev2 = kNegInf;
  if (j1 + 1 < n) {
    pm2_next1 = pm2_r[row_i + j1 + 1];  // Pm2(i, j+2)
    w_next1 = w_r[row_i + j1 + 1];      // W(i, j+2)
  }
  if (i >= 1 && has1) sp_prev1 = sp_c[col_j + ld + i - 1];  // prefix of column j+1 up to row i-1
  if (i >= 2) sp_prev2 = sp_c[col_j + i - 2];               // prefix of column j up to row i-2
  float qa0 = kNegInf, mbc0 = 0.f, qa1 = kNegInf, mbc1 = 0.f;
  const float ztot = q.zp[n], zpi = q.zp[i];
  float zsj0 = 0.f, zsj1 = 0.f;
  if (paired0) {
    qa0 = q.m[T_QA][row_i + j];
    mbc0 = model.mbclose(s, n, i, j);
    zsj0 = q.zs[j + 1];
  }
  if (paired1) {
    qa1 = q.m[T_QA][row_i + j1];
    mbc1 = model.mbclose(s, n, i, j1);
    zsj1 = q.zs[j1 + 1];
  }
  Acc acc[NA];
#pragma unroll
  for (int x = 0; x < NA; x++) acc[x] = acc_empty();
  // [0] Pm(i,j), [1] Pm(i-1,j), [2] Pm(i,j+1): k = j+1 .. n-1 (the k = j+1 term of [0] reads
  // W(i,j+1) of this launch next to Q1(j+1,j) = -inf: masked like [2]'s)
#ifdef RNAMC_DEBUG_KNOBS
  if (!(b.debug & 2))
#endif
  {
    if (has1 && n - 1 - j >= 2)
      acc_product_3<TPC>(acc[0], acc[1], acc[2], w_r + row_i + j1, w_r + row_i - ld + j1,
                         q.m[T_Q1R] + static_cast<size_t>(j1) * ld + j,
                         q.m[T_Q1R] + static_cast<size_t>(j1 + 1) * ld + j, n - 1 - j, i >= 1, true, t);
    // [5] [6] L_e cases one and three of (i,j) and (i,j+1): k = 0 .. i-1, Q1(k+1,i-1) + R(k,.)
    if (i >= 1 && (paired0 || paired1)) {
      if (has1)
        acc_product_2b<TPC>(acc[5], acc[6], q.m[T_Q1C] + static_cast<size_t>(i - 1) * ld + 1, r_c + col_j,
                            r_c + col_j + ld, i - 1, i, t);
      else
        acc_product<TPC>(acc[5], q.m[T_Q1C] + static_cast<size_t>(i - 1) * ld + 1, r_c + col_j, i - 1, t);
    }
  }
  // [3] [4] enclosing 2-loops
  if (paired0) outer_block<CONTRA, TPC>(b, q, acc[3], i, j, t, qb0);
  if (paired1) outer_block<CONTRA, TPC>(b, q, acc[4], i, j1, t, qb1);
  if (!cell_reduce<NA, TPC>(acc, red)) return;

  const bool st = t == 0u;
  const float mb_un = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  const float ext_bp = CONTRA ? b.params->contra.external_score_basepair : 0.f;
  const float abr = CONTRA ? b.params->contra.multibranch_score_basepair
                           : b.params->turner.coeff_num_branches;
  // ---- cell (i,j+1)
  float w1 = kNegInf, pm2_1 = kNegInf;
  if (has1) {
    const float pm1 = acc_value(acc[2]);
    pm2_1 = lse2(pm2_next1 + mb_un, w_next1);
    if (st) {
      pm2_r[row_i + j1] = pm2_1;
      r_c[col_j + ld + i] = lse2(pm1, pm2_1);
      sp_c[col_j + ld + i] = lse2(sp_prev1 + mb_un, pm1);
    }
    if (paired1) {
      const float ext = CONTRA ? (((zpi + zsj1) + qa1) + ext_bp) - ztot : ((zpi + qa1) + zsj1) - ztot;
      Acc pa = acc[4];
      acc_add(pa, ext);
      const float A = qa1 + abr;
      acc_add(pa, A + acc_value(acc[6]));
      acc_add(pa, A + sp_prev1);
      const float lp = acc_value(pa);
      if (lp > kNegInf) {
        w1 = (lp + mbc1) - qb1;
        if (st) {
          q.out[tri_off(n, d + 1) + i] = lp;
          q.m[T_ZRE][row_i + j1] = w1;
          reinterpret_cast<float2*>(q.m[T_PQ])[row_i + j1] = make_float2(lp, qb1);
        }
      }
    }
  }
  // ---- cell (i,j); the prefix of column j up to row i-1 needs Pm(i-1,j) of this launch
  const float pmn = acc_value(acc[1]);
  const float sp_prev0 = i >= 1 ? lse2(sp_prev2 + mb_un, pmn) : kNegInf;
  const float pm0 = acc_value(acc[0]);
  const float pm2_0 = lse2(pm2_1 + mb_un, w1);
  if (st) {
    pm2_r[row_i + j] = pm2_0;
    r_c[col_j + i] = lse2(pm0, pm2_0);
    sp_c[col_j + i] = lse2(sp_prev0 + mb_un, pm0);
  }
  if (!paired0) return;
  const float ext = CONTRA ? (((zpi + zsj0) + qa0) + ext_bp) - ztot : ((zpi + qa0) + zsj0) - ztot;
  Acc pa = acc[3];
  acc_add(pa, ext);
  const float A = qa0 + abr;
  acc_add(pa, A + acc_value(acc[5]));
  acc_add(pa, A + sp_prev0);
  const float lp = acc_value(pa);
  if (st && lp > kNegInf) {
    q.out[tri_off(n, d) + i] = lp;
    q.m[T_ZRE][row_i + j] = (lp + mbc0) - qb0;
    reinterpret_cast<float2*>(q.m[T_PQ])[row_i + j] = make_float2(lp, qb0);
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

// threads per cell by the number of cells of the diagonal (all sequences of the group): many
// cells -> one wave each (four cells per workgroup, no barrier); few cells with long sums ->
// 1024 threads each
#define RNAMC_TREE_LAUNCH(K, C, T)                                                              \
  hipLaunchKernelGGL((K<C, T>), dim3((cells + (T < 256 ? 256 / T : 1) - 1) / (T < 256 ? 256 / T : 1), nseq, 1), \
                     dim3(T < 256 ? 256 : T), 0, st, b, d)
static int tree_tpc(uint64_t cells, int64_t knob) {
  if (knob == 64 || knob == 256 || knob == 1024) return static_cast<int>(knob);
  return cells >= 2048 ? 64 : (cells >= 192 ? 256 : 1024);
}

void launch_tree_inside(const TreeBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                        int64_t tpc_knob, bool two, hipStream_t st) {
  const uint32_t cells = max_n - d;
  const int tpc = tree_tpc(static_cast<uint64_t>(cells) * nseq, tpc_knob);
  if (two) {
    if (contra) {
      if (tpc == 64) RNAMC_TREE_LAUNCH(k_tree_inside2, true, 64);
      else if (tpc == 256) RNAMC_TREE_LAUNCH(k_tree_inside2, true, 256);
      else RNAMC_TREE_LAUNCH(k_tree_inside2, true, 1024);
    } else {
      if (tpc == 64) RNAMC_TREE_LAUNCH(k_tree_inside2, false, 64);
      else if (tpc == 256) RNAMC_TREE_LAUNCH(k_tree_inside2, false, 256);
      else RNAMC_TREE_LAUNCH(k_tree_inside2, false, 1024);
    }
    return;
  }
  if (contra) {
    if (tpc == 64) RNAMC_TREE_LAUNCH(k_tree_inside, true, 64);
    else if (tpc == 256) RNAMC_TREE_LAUNCH(k_tree_inside, true, 256);
    else RNAMC_TREE_LAUNCH(k_tree_inside, true, 1024);
  } else {
    if (tpc == 64) RNAMC_TREE_LAUNCH(k_tree_inside, false, 64);
    else if (tpc == 256) RNAMC_TREE_LAUNCH(k_tree_inside, false, 256);
    else RNAMC_TREE_LAUNCH(k_tree_inside, false, 1024);
  }
}

void launch_tree_outside(const TreeBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                         int64_t tpc_knob, bool two, hipStream_t st) {
  const uint32_t cells = max_n - d;
  const int tpc = tree_tpc(static_cast<uint64_t>(cells) * nseq, tpc_knob);
  if (two) {
    if (contra) {
      if (tpc == 64) RNAMC_TREE_LAUNCH(k_tree_outside2, true, 64);
      else if (tpc == 256) RNAMC_TREE_LAUNCH(k_tree_outside2, true, 256);
      else RNAMC_TREE_LAUNCH(k_tree_outside2, true, 1024);
    } else {
      if (tpc == 64) RNAMC_TREE_LAUNCH(k_tree_outside2, false, 64);
      else if (tpc == 256) RNAMC_TREE_LAUNCH(k_tree_outside2, false, 256);
      else RNAMC_TREE_LAUNCH(k_tree_outside2, false, 1024);
    }
    return;
  }
  if (contra) {
    if (tpc == 64) RNAMC_TREE_LAUNCH(k_tree_outside, true, 64);
    else if (tpc == 256) RNAMC_TREE_LAUNCH(k_tree_outside, true, 256);
    else RNAMC_TREE_LAUNCH(k_tree_outside, true, 1024);
  } else {
    if (tpc == 64) RNAMC_TREE_LAUNCH(k_tree_outside, false, 64);
    else if (tpc == 256) RNAMC_TREE_LAUNCH(k_tree_outside, false, 256);
    else RNAMC_TREE_LAUNCH(k_tree_outside, false, 1024);
  }
#undef RNAMC_TREE_LAUNCH
}

void launch_tree_finalize(const TreeBatch& b, uint32_t nseq, uint32_t max_n, hipStream_t st) {
  const uint64_t elems = static_cast<uint64_t>(max_n) * (max_n + 1) / 2;
  uint32_t gx = static_cast<uint32_t>(std::min<uint64_t>((elems + 1023) / 1024, 1024));
  if (gx == 0) gx = 1;
  hipLaunchKernelGGL(k_tree_finalize, dim3(gx, nseq, 1), dim3(256), 0, st, b);
}

}  // namespace rnamc
