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
//  * one lane owns one cell (i, i+d) — two cells (i, i+d), (i, i+d+1) in the large inside
//    launches, which fold both off one stream of the shared row operands; lanes of a wave
//    own consecutive i, so with the packed layouts of rnamc_internal.h the operands of the
//    inside folds and of probs_multibranch are contiguous 256-B wave accesses, and the
//    pair-probability tail streams lane-private columns in whole 128-byte lines;
//  * the reduction index k is walked sequentially per lane (order is part of the result);
//    parallelism comes from cells x sequences x independent folds, not from k.  Operands
//    of the next chunk of k-steps are fetched while the current chunk is folded;
//  * the 8-piece cubic of logsumexp is evaluated branch-free: a 42-cell LDS table
//    and one compare give the piece, one ds_read_b128 its 4 coefficients;
//  * hot loops carry no exec-mask branches: loads are unconditional (matrices are
//    padded), lane validity is applied to the loaded values; wave-uniform trip counts are
//    passed through readfirstlane so that loop control stays scalar;
//  * registers are spent on occupancy rather than on deep software pipelines (the sweeps
//    are bound by the loads the resident waves keep in flight): the pair tail holds ONE
//    line per stream, and runs as a kernel of its own beside the other outside roles;
//  * each launch carries independent roles in disjoint blocks: inside = {folds of
//    diagonals d, d+1; early part of the closing-pair blocks of d+2, d+3}, outside =
//    {probs_multibranch of d, 2-loop half of the pair probabilities of d-1} beside
//    {multibranch half of the pair probabilities of d};
//  * block ids walk over sequences first so that the 8 XCDs get equal mixes of
//    light and heavy blocks; heavy blocks are issued first.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <type_traits>

#include "rnamc_device.h"
#include "rnamc_scoring.h"

namespace rnamc {

namespace {

constexpr int kU = RNAMC_KU;  // k-steps fetched ahead per lane

// ----------------------------------------------------------------------------
// numerics: src/utils.rs:579-655

// ln_exp_1p (src/utils.rs:602-627) is an 8-piece cubic; piece 8 below is the
// identity (0,0,1,0): ((0*z+0)*z+1)*z+0 == z exactly, which is what logsumexp
// adds when z >= LOGSUMEXP_THRESHOLD_UPPER (src/utils.rs:589-591).
constexpr int kLsePieces = 9;
__constant__ float kLseCoef[kLsePieces][4] = {
    {-0.0065591595f, 0.12764427f, 0.49965546f, 0.6931542f},
    {-0.015515756f, 0.14467756f, 0.48829398f, 0.6958093f},
    {-0.012890925f, 0.13010283f, 0.51503986f, 0.6795586f},
    {-0.0072142647f, 0.087754086f, 0.6208708f, 0.5909676f},
    {-0.0031455354f, 0.046722945f, 0.7592532f, 0.43487945f},
    {-0.0010110698f, 0.018594341f, 0.88317305f, 0.25236955f},
    {-0.000196278f, 0.0046084408f, 0.9634432f, 0.09831489f},
    {-0.0000113994f, 0.0003734731f, 0.9959107f, 0.0149855051f},
    {0.f, 0.f, 1.f, 0.f},
};
// upper bounds of pieces 0..7 (`x < t` picks the lower piece)
__constant__ float kLseBreaks[8] = {0.66153675f, 1.6320158f, 2.4912589f, 3.3792500f,
                                    4.426169f,   5.789071f,  7.8162727f, 11.862479f};

// Piece lookup without a compare ladder: the top 12 bits of z (sign, exponent,
// 3 mantissa bits) select one of 42 cells covering [0, 0.5), the 40 eighth-binade
// cells of [0.5, 16) and [16, inf); no cell holds more than one breakpoint, so a
// cell entry {breakpoint or +inf, byte offset of the lower piece} and ONE compare
// give the piece.  LDS image: 9 coefficient rows (144 B) then 42 cell entries.
constexpr int kLseCells = 42;
constexpr int kLseCellLo = 0x3F0 - 1;  // (bits(0.5f) >> 20) - 1
constexpr int kLseCellHi = 0x418;      //  bits(16.f) >> 20
struct LseTab {
  float4 coef[kLsePieces];
  float2 cell[kLseCells];  // {threshold, byte offset of lower piece as float bits}
};

__device__ __forceinline__ void load_lse_table(LseTab* tab) {
  const uint32_t t = threadIdx.x;
  if (t < kLsePieces)
    tab->coef[t] = make_float4(kLseCoef[t][0], kLseCoef[t][1], kLseCoef[t][2], kLseCoef[t][3]);
  if (t < kLseCells) {
    // cell t covers [lo, hi)
    float lo, hi;
    if (t == 0) {
      lo = 0.f, hi = 0.5f;
    } else if (t == kLseCells - 1) {
      lo = 16.f, hi = __builtin_inff();
    } else {
      lo = __uint_as_float(static_cast<uint32_t>(kLseCellLo + static_cast<int>(t)) << 20);
      hi = __uint_as_float(static_cast<uint32_t>(kLseCellLo + static_cast<int>(t) + 1) << 20);
    }
    // piece of `lo`: number of breakpoints <= lo
    int piece = 0;
    for (int x = 0; x < 8; x++) piece += (lo >= kLseBreaks[x]) ? 1 : 0;
    float thr = __builtin_inff();
    if (piece < 8 && kLseBreaks[piece] < hi) thr = kLseBreaks[piece];
    tab->cell[t] = make_float2(thr, __uint_as_float(static_cast<uint32_t>(piece) * 16u));
  }
  __syncthreads();
}

// v_max_f32 / v_min_f32 without the sNaN-quieting canonicalisation hipcc adds in
// front of fmaxf/fminf (operands here are finite or -inf, never NaN).
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// One fold step sum ⊕ x.  Operands are finite or -inf (never NaN/+inf: absent
// map entries are -inf and are masked at the source), so the reference's two
// is_finite() early-outs collapse to "if the smaller one is -inf take the
// larger one".  Piece boundaries: `x < t` in the reference <=> !(z >= t).
__device__ __forceinline__ float lse(float sum, float x, const LseTab* tab) {
  const float hi = vmax(sum, x);
  const float lo = vmin(sum, x);
  const float z = hi - lo;  // >= 0, or +inf / NaN when lo is -inf (result then discarded)
  const int e = static_cast<int>(__float_as_uint(z) >> 20);
  const int cell = min(max(e, kLseCellLo), kLseCellHi) - kLseCellLo;  // v_med3_i32
  const float2 ce = tab->cell[cell];
  const uint32_t boff = __float_as_uint(ce.y) + ((z >= ce.x) ? 16u : 0u);
  const float4 co = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tab) + boff);
  float r = ((co.x * z + co.y) * z + co.z) * z + co.w;
  r = lo + r;
  return (lo == kNegInf) ? hi : r;
}

// The same fold step with a WAVE-UNIFORM fast path for the outside chains.  Measured on the 512
// longest sequences of the bench batch (profiles/r04_wave_uniform_far.txt): in 62 % of the fold
// steps of the pair-probability tail and 64 % of those of probs_multibranch (53 % under
// CONTRAfold) EVERY lane of the wave either meets the identity piece (z >= 11.862479,
// src/utils.rs:589-591) or folds a -inf operand.  One compare + a scalar branch on its ballot
// then replaces the two dependent LDS lookups and the cubic by the reference's own two operations
// `lo + (hi - lo)`; lanes with lo = -inf take hi (src/utils.rs:581-586).  Bit-identical by
// construction: piece 8 of the table evaluates ((0*z+0)*z+1)*z+0 = z.  (z is NaN only when both
// operands are -inf: `z < t` is false, the lane counts as far, and the select returns hi = -inf.)
#ifndef RNAMC_LSE_WU
#define RNAMC_LSE_WU 1
#endif
__device__ __forceinline__ float lse_wu(float sum, float x, const LseTab* tab) {
#if RNAMC_LSE_WU
  const float hi = vmax(sum, x);
  const float lo = vmin(sum, x);
  const float z = hi - lo;
  if (__ballot(z < 11.862479f) == 0ull) {
    const float r = lo + z;
    return (lo == kNegInf) ? hi : r;
  }
  const int e = static_cast<int>(__float_as_uint(z) >> 20);
  const int cell = min(max(e, kLseCellLo), kLseCellHi) - kLseCellLo;
  const float2 ce = tab->cell[cell];
  const uint32_t boff = __float_as_uint(ce.y) + ((z >= ce.x) ? 16u : 0u);
  const float4 co = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tab) + boff);
  float r = ((co.x * z + co.y) * z + co.z) * z + co.w;
  r = lo + r;
  return (lo == kNegInf) ? hi : r;
#else
  return lse(sum, x, tab);
#endif
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

#ifdef RNAMC_COUNT_FAR
// Counting build only (make COUNT_FAR=1 -> librnamc_count.so, scripts/count_far.py): how often a
// whole WAVE of the outside chains meets the identity piece of logsumexp (z >= 11.862479, or the
// smaller operand -inf) in the same fold step — the condition under which a wave-uniform fast
// path could skip the table lookups and the cubic.  Never part of the product library.
//   [role][0] fold steps executed by a wave     [1] ... with every lane on the identity piece / -inf
//   [2] ... with every lane's TERM -inf (no-op)  [3] lane-steps on the identity piece / -inf
//   [4] lane-steps whose term is -inf            [5] k-steps (all folds of one k) executed by a wave
//   [6] ... all of them wave-uniform far         [7] chunks of 16 k (pair tail) / 8 k (mb) [8] ... all far
// role 0 = pair tail (k_outside<.,2> / the tail role of <.,7>), 1 = probs_multibranch
__device__ unsigned long long g_far_counters[2][16];
struct FarCount {
  uint32_t c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool kfar = true, cfar = true;
  __device__ __forceinline__ void fold(float sum, float x) {
    const float hi = fmaxf(sum, x), lo = fminf(sum, x);
    const float z = hi - lo;
    const bool far = !(z < 11.862479f);  // identity piece; lo = -inf gives z = +inf or NaN
    const bool skip = !(x > kNegInf);
    const unsigned long long all = __ballot(true);
    const unsigned long long mf = __ballot(far), ms = __ballot(skip);
    c[0]++;
    c[1] += (mf == all) ? 1u : 0u;
    c[2] += (ms == all) ? 1u : 0u;
    c[3] += static_cast<uint32_t>(__popcll(mf));
    c[4] += static_cast<uint32_t>(__popcll(ms));
    kfar = kfar && (mf == all);
  }
  __device__ __forceinline__ void end_k() {
    c[5]++;
    c[6] += kfar ? 1u : 0u;
    cfar = cfar && kfar;
    kfar = true;
  }
  __device__ __forceinline__ void end_chunk() {
    c[7]++;
    c[8] += cfar ? 1u : 0u;
    cfar = true;
  }
  __device__ __forceinline__ void flush(int role) {
    if ((threadIdx.x & 63u) == 0u)
      for (int x = 0; x < 9; x++) atomicAdd(&g_far_counters[role][x], static_cast<unsigned long long>(c[x]));
  }
};
#endif

// ----------------------------------------------------------------------------
// index algebra

// load ubase[i] as global_load_dword v, v_off, s[base] (uniform 64-bit base in
// SGPRs + the lane's 32-bit byte offset) instead of a 64-bit VALU address per load
__device__ __forceinline__ float ldu(const float* __restrict__ ubase, uint32_t lane_byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ubase) + lane_byte_off);
}

// Two-stage software pipeline over `nchunks` whole chunks of kU steps starting at t:
// fetch(buf, t) loads the operands of steps t..t+kU-1 into a register buffer,
// compute(buf, t) folds them.  The loads of chunk c+1 are issued before chunk c is
// folded and land in the OTHER buffer, so their latency hides under ~kU fold steps.
template <class Buf, int U = kU, bool ALWAYS_FETCH = false, class Fetch, class Compute>
__device__ __forceinline__ uint32_t pingpong(uint32_t t, uint32_t nchunks, Fetch&& fetch,
                                             Compute&& compute) {
  if (nchunks == 0) return t;
  Buf A, B;
  fetch(A, t);
  uint32_t c = 0;
  // ALWAYS_FETCH: the next chunk is fetched unconditionally (the last iteration re-fetches
  // its own chunk).  A conditional fetch makes the compiler's s_waitcnt placement assume
  // the no-fetch path; which form is faster is measured per call site.
  for (;;) {
    if (ALWAYS_FETCH) {
      fetch(B, (c + 1 < nchunks) ? t + U : t);
    } else if (c + 1 < nchunks) {
      fetch(B, t + U);
    }
    compute(A, t);
    t += U;
    c++;
    if (c >= nchunks) break;
    if (ALWAYS_FETCH) {
      fetch(A, (c + 1 < nchunks) ? t + U : t);
    } else if (c + 1 < nchunks) {
      fetch(A, t + U);
    }
    compute(B, t);
    t += U;
    c++;
    if (c >= nchunks) break;
  }
  return t;
}

__device__ __forceinline__ uint32_t tri_off(uint32_t n, uint32_t d) {
  // start of diagonal d (diag-major) == start of row d (row-major)
  return d * n - (d * (d - 1u)) / 2u;
}

// Column-major packed triangle with every column padded to a multiple of 16 f32:
// element (k, j), k <= j, at col_off(j) + k.  A lane that walks one column reads
// 64-byte-aligned blocks of its own, so a walk over k fetches exactly the bytes it
// uses (the multibranch half of the pair probabilities walks columns of
// probs_multibranch{,2} and of sums_1ormore_basepairs).
__device__ __forceinline__ uint32_t col_off(uint32_t j) {
  const uint32_t m = j >> 4, r = j & 15u;
  return 16u * (m + 1u) * (8u * m + r);
}

struct Seq {
  const uint8_t* s;  // base codes
  uint32_t n;
  float* m[M_COUNT];
  float* out;          // packed bpp triangle (log domain until finalize)
  const uint32_t* pk;  // 2-bit packed bases, 16 per word, position p at bit 2(p+32)
  // canonical-pair cells of every diagonal, ascending i, at tri_off(n,d); counts per d
  uint16_t* cidx;
  uint32_t* ccnt;
  // c64[w * (n + 64) + D]: canonical cells of diagonal D at positions < 64 w (0 for D >= n)
  uint32_t* c64;
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
  q.pk = reinterpret_cast<const uint32_t*>(b.workspace + sd.pk_off);
  q.cidx = reinterpret_cast<uint16_t*>(b.workspace + sd.cidx_off);
  q.ccnt = reinterpret_cast<uint32_t*>(b.workspace + sd.ccnt_off);
  q.c64 = reinterpret_cast<uint32_t*>(b.workspace + sd.c64_off);
  return q;
}

// 32 consecutive bases in two registers: window position q holds base p0+q.
struct Win {
  uint32_t lo, hi;
};

// p0 may be as low as -32 (the packed copy carries 32 zero bases in front and
// >= 64 behind, written by k_init).
__device__ __forceinline__ Win load_win(const uint32_t* __restrict__ pk, int p0) {
  const uint32_t bit = 2u * static_cast<uint32_t>(p0 + 32);
  const uint32_t w = bit >> 5, sh = bit & 31u;
  const uint32_t w0 = pk[w], w1 = pk[w + 1], w2 = pk[w + 2];
  Win r;
  r.lo = __builtin_amdgcn_alignbit(w1, w0, sh);
  r.hi = __builtin_amdgcn_alignbit(w2, w1, sh);
  return r;
}

// base at window position q; q is wave-uniform at every call site
__device__ __forceinline__ int wbase(const Win& w, uint32_t q) {
  const uint32_t h = (q < 16u) ? w.lo : w.hi;
  return static_cast<int>((h >> (2u * (q & 15u))) & 3u);
}

__device__ __forceinline__ int idx4(int a, int b, int c, int d) { return ((a * 4 + b) * 4 + c) * 4 + d; }

// base codes around one 2-loop: "close" = the closing pair (c0,c1) with its inner
// neighbours x1 = s[c0+1], y1 = s[c1-1], x2 = s[c0+2], y2 = s[c1-2]; "inner" = the
// enclosed pair (a0,a1) with its outer neighbours o0 = s[a0-1], o1 = s[a1+1].
struct TwoLoopCodes {
  int c0, c1, x1, y1, x2, y2;  // closing side
  int a0, a1, o0, o1;          // enclosed side
};

constexpr int kPU = RNAMC_KPU;                          // probes fetched ahead
constexpr uint32_t kProbes = (RNAMC_MAX_2LOOP_LEN + 1) * (RNAMC_MAX_2LOOP_LEN + 2) / 2;  // 496
static_assert(kProbes % kPU == 0, "probe list is walked in whole chunks");
static_assert(RNAMC_MAX_2LOOP_LEN == RNAMC_MAX_LOOP_LEN, "one probe triangle for both models");

#include "rnamc_probes.h"

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
  // 2-bit packed copy of the sequence: base p at bit 2(p+32); zeros around it
  uint32_t* pk = reinterpret_cast<uint32_t*>(b.workspace + sd.pk_off);
  const uint8_t* s = b.bases + sd.seq_off;
  for (size_t wd = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; wd < sd.pk_words;
       wd += stride) {
    uint32_t v = 0;
    for (uint32_t y = 0; y < 16; y++) {
      const int64_t pos = static_cast<int64_t>(wd) * 16 + y - 32;
      if (pos >= 0 && pos < static_cast<int64_t>(sd.n)) v |= static_cast<uint32_t>(s[pos] & 3u) << (2u * y);
    }
    pk[wd] = v;
  }
}

// Lists of the cells that can hold a pair: for every diagonal d the ascending i
// with a canonical (s[i], s[i+d]).  The closing-pair block and the outside pair
// block run one lane per LISTED cell, so no lane idles on the ~62 % of cells that
// can never pair.  One wave per (sequence, diagonal).
__global__ void __launch_bounds__(64) k_compact(DeviceBatch b) {
  const Seq q = load_seq(b, blockIdx.y);
  const uint32_t n = q.n;
  const uint32_t d = blockIdx.x;
  if (d >= n) return;
  const uint8_t* s = q.s;
  uint16_t* dst = q.cidx + tri_off(n, d);
  const uint32_t lane = threadIdx.x;
  const uint32_t stride = n + 64u, nb64 = (n + 63u) / 64u;
  uint32_t cnt = 0;
  for (uint32_t i0 = 0; i0 < n - d; i0 += 64) {
    const uint32_t i = i0 + lane;
    const bool c = (i < n - d) && canonical(s[i], s[i + d]);
    const unsigned long long m = __ballot(c);
    if (lane == 0) q.c64[(i0 >> 6) * stride + d] = cnt;
    if (c) dst[cnt + __popcll(m & ((1ull << lane) - 1ull))] = static_cast<uint16_t>(i);
    cnt += static_cast<uint32_t>(__popcll(m));
  }
  if (lane == 0) q.ccnt[d] = cnt;
  // 64-blocks past the end of this diagonal, and the columns D >= n that a chunked walk
  // may touch: every entry of the table is defined
  for (uint32_t w = (n - d + 63u) / 64u + lane; w < nb64; w += 64) q.c64[w * stride + d] = cnt;
  if (d == 0)
    for (uint32_t x = lane; x < nb64 * 64u; x += 64) q.c64[(x >> 6) * stride + n + (x & 63u)] = 0u;
}

// lane -> listed cell of diagonal d; returns false for lanes past the list
__device__ __forceinline__ bool listed_cell(const Seq& q, uint32_t d, uint32_t t, uint32_t cnt,
                                            uint32_t& i) {
  const bool valid = t < cnt;
  i = valid ? static_cast<uint32_t>(q.cidx[tri_off(q.n, d) + t]) : 0u;
  return valid;
}

// ----------------------------------------------------------------------------
// inside pass, closing-pair block of one cell of diagonal d
// (src/mccaskill_algo.rs:297-343 Turner, 400-467 CONTRAfold)
// The block is a left fold: hairpin, the <= 496 enclosed pairs, then the multibranch
// term.  Everything but the last term needs only sums_close of spans <= d-2, so the
// two-diagonal schedule evaluates that part early (PAIR_HEAD: the running sum is parked
// in the sums_close slot) beside the folds of older diagonals, and adds the multibranch
// term once sums_multibranch of diagonal d-2 exists (PAIR_TAIL).  PAIR_FULL does both.
enum PairMode { PAIR_FULL = 0, PAIR_HEAD = 1, PAIR_TAIL = 2 };

// LDS image of the probe tables only in kernels that fold probes (the fold-only kernels keep
// their LDS footprint small).
struct NoProbeTabs {
  float unused;
};
template <bool WANT>
using ProbeTabStore = typename std::conditional<WANT, ProbeTabs, NoProbeTabs>::type;

template <bool CONTRA, int MODE>
__device__ __forceinline__ void inside_pair_cell(const DeviceBatch& b, const Seq& q, uint32_t d,
                                                 uint32_t i, bool valid, const LseTab* tab,
                                                 const ProbeTabs& L) {
  const uint32_t n = q.n;
  const uint32_t j = i + d;
  const uint8_t* s = q.s;
  bool act = valid && canonical(s[i], s[j]);
  if (!(b.allows_short_hairpins && CONTRA) && d + 1 < RNAMC_MIN_SPAN_HAIRPIN_CLOSE) act = false;
  if (__ballot(act) == 0ull) return;
  const auto model = ModelOf<CONTRA>::make(b);
  const uint32_t o = tri_off(n, d) + i;

  float sum = kNegInf;
  if (MODE != PAIR_TAIL) {
    if (act && (!CONTRA || d - 1 <= RNAMC_MAX_LOOP_LEN))
      sum = lse(sum, model.hairpin(s, n, i, j), tab);
    // enclosed pairs (k,l) = (i+1+a, j-1-bb), a ascending, bb ascending (l descending),
    // a+bb <= 30, k < j-1, l > k   <=>   a + bb <= d-3   (uniform over the diagonal)
    if (d >= 3) {
      const uint32_t lim = min(static_cast<uint32_t>(RNAMC_MAX_2LOOP_LEN), d - 3);
      sum = probe_fold<CONTRA, false>(b, q, d, i, act, lim, sum, 0.f, tab, L);
    }
    if (!act) return;
    if (MODE == PAIR_HEAD) {
      q.m[M_QB][o] = sum;
      return;
    }
  } else {
    if (!act) return;
    sum = q.m[M_QB][o];
  }
  const float mbc = model.mbclose(s, n, i, j);
  const float qm = (d >= 2) ? q.m[M_QM][tri_off(n, d - 2) + i + 1] : kNegInf;
  sum = lse(sum, qm + mbc, tab);
  const float acc = model.accessible(s, n, i, j);
  if (sum > kNegInf) {
    q.m[M_MBC][o] = mbc;
    q.m[M_QB][o] = sum;
    q.m[M_QA][o] = sum + acc;
  }
}

// ----------------------------------------------------------------------------
// inside pass, the Theta(n) folds of one cell of diagonal d
// (src/mccaskill_algo.rs:344-374 Turner, 468-512 CONTRAfold)
template <bool CONTRA>
__device__ __forceinline__ void inside_sums_cell(const DeviceBatch& b, const Seq& q, uint32_t d,
                                                 uint32_t i, const LseTab* tab) {
  const uint32_t n = q.n;
  const uint32_t od = tri_off(n, d) + i;  // this cell, diag-major
  const float* __restrict__ zre = q.m[M_ZRE];
  const float* __restrict__ zrm = q.m[CONTRA ? M_ZRM : M_ZRE];
  const float* __restrict__ z = q.m[M_Z];
  const float* __restrict__ q1 = q.m[M_Q1D];

  float zr_ext, zr_mb;
  float c = 0.f, mun = 0.f;
  if (!CONTRA) {
    // sums_rightmost_basepairs_external(i,j) = fold_{k=i+1..j} sums_accessible(i,k); the
    // terms do not depend on j, so the fold of (i,j-1) extended by one step is the
    // same sequence of operations (344-351).
    const float prev = (d >= 1) ? zre[tri_off(n, d - 1) + i] : kNegInf;
    zr_ext = lse(prev, q.m[M_QA][od], tab);
    zr_mb = zr_ext;
    q.m[M_ZRE][od] = zr_ext;
    c = b.params->turner.coeff_num_branches;
  } else {
    const rnamc_fold_score_sets& f = b.params->contra;
    const float ebp = f.external_score_basepair, eun = f.external_score_unpair;
    const float mbp = f.multibranch_score_basepair;
    mun = f.multibranch_score_unpair;
    const float* __restrict__ qa = q.m[M_QA];
    zr_ext = kNegInf;
    zr_mb = kNegInf;
    // k = i + t, j - k = d - t; sums_accessible(i,k) is -inf when (i,k) is no pair
    struct ABuf {
      float xs[kU];
    };
    uint32_t t = pingpong<ABuf>(
        1u, d / kU,
        [&](ABuf& B, uint32_t t0) {
#pragma unroll
          for (int u = 0; u < kU; u++) B.xs[u] = ldu(qa + tri_off(n, t0 + u), i * 4u);
        },
        [&](const ABuf& B, uint32_t t0) {
#pragma unroll
          for (int u = 0; u < kU; u++) {
            const float cnt = static_cast<float>(d - t0 - u);
            zr_ext = lse(zr_ext, B.xs[u] + ebp + eun * cnt, tab);
            zr_mb = lse(zr_mb, B.xs[u] + mbp + mun * cnt, tab);
          }
        });
    for (; t <= d; t++) {
      const float x = qa[tri_off(n, t) + i];
      const float cnt = static_cast<float>(d - t);
      zr_ext = lse(zr_ext, x + ebp + eun * cnt, tab);
      zr_mb = lse(zr_mb, x + mbp + mun * cnt, tab);
    }
    q.m[M_ZRE][od] = zr_ext;
    q.m[M_ZRM][od] = zr_mb;
  }

  // sums_external (352-363 / 487-498) and sums_1ormore / sums_multibranch
  // (364-374 / 499-512) share the walk k = i + t:
  //   Zr[k][j]   -> diagonal d-t, offset i+t      Z/Q1[i][k-1] -> diagonal t-1, offset i
  float ext, s1, s2 = kNegInf;
  if (!CONTRA) {
    ext = lse(0.f, zr_ext + 0.f, tab);  // k = i: Z[i][i-1] is the lower-triangle 0
    s1 = zr_ext + c;
  } else {
    ext = lse(b.params->contra.external_score_unpair * static_cast<float>(d + 1), zr_ext + 0.f, tab);
    s1 = zr_mb;
  }
  auto step = [&](float re, float rm, float zz, float qq, uint32_t t) {
    ext = lse(ext, re + zz, tab);
    if (!CONTRA) {
      const float x = re + c;
      s1 = lse(s1, x, tab);
      s2 = lse(s2, qq + x, tab);
    } else {
      s1 = lse(s1, rm + mun * static_cast<float>(t), tab);
      s2 = lse(s2, qq + rm, tab);
    }
  };
  const uint32_t i4 = i * 4u;
  struct SBuf {
    float re[kU], rm[kU], zz[kU], qq[kU];
  };
  // steps 1 .. d-1
  uint32_t t = pingpong<SBuf, kU, true>(
      1u, d >= 1 ? (d - 1) / kU : 0u,
      [&](SBuf& B, uint32_t t0) {
#pragma unroll
        for (int u = 0; u < kU; u++) {
          const uint32_t orr = tri_off(n, d - t0 - u) + t0 + u;
          const uint32_t o = tri_off(n, t0 + u - 1);
          B.re[u] = ldu(zre + orr, i4);
          B.rm[u] = CONTRA ? ldu(zrm + orr, i4) : 0.f;
          B.zz[u] = ldu(z + o, i4);
          B.qq[u] = ldu(q1 + o, i4);
        }
      },
      [&](const SBuf& B, uint32_t t0) {
#pragma unroll
        for (int u = 0; u < kU; u++) step(B.re[u], B.rm[u], B.zz[u], B.qq[u], t0 + u);
      });
  for (; t < d; t++) {
    const uint32_t orr = tri_off(n, d - t) + t + i;
    const uint32_t o = tri_off(n, t - 1) + i;
    step(zre[orr], CONTRA ? zrm[orr] : 0.f, z[o], q1[o], t);
  }
  q.m[M_Z][od] = ext;
  q.m[M_QM][od] = s2;
  s1 = lse(s1, s2, tab);
  q.m[M_Q1D][od] = s1;
  if (i >= 1) q.m[M_Q1C][col_off(i + d) + i - 1] = s1;  // column j, shifted one row up
}

// Two diagonals per lane: the folds of cells (i, i+d) and (i, i+d+1) walk the same row
// operands sums_external(i,k-1) / sums_1ormore(i,k-1), so one lane folds both cells off
// one stream of them (4 loads per 6 fold steps instead of 6; the second cell's column
// operand Zr(k, j+1) is the neighbouring lane's first-column operand of the step before:
// same cache lines).  Needs the closing-pair blocks of diagonals d and d+1 done
// (sums_multibranch of diagonals <= d-1) and every fold of diagonals <= d-1.  Same
// operations per cell, in the same order, as inside_sums_cell.
//
// Turner: the rightmost-pair sums are one fold step from final values; Zr(i+1, j+1), which
// a neighbouring lane produces in this very launch, is recomputed here.
// CONTRAfold: they are folds of their own (inside_zr_pair2, the launch before this one).

// CONTRAfold: sums_rightmost_basepairs_{external,multibranch} of cells (i, i+d) and
// (i, i+d+1) off one stream of sums_accessible(i, k) (src/mccaskill_algo.rs:468-486)
__device__ __forceinline__ void inside_zr_pair2(const DeviceBatch& b, const Seq& q, uint32_t d,
                                                uint32_t i, const LseTab* tab) {
  const uint32_t n = q.n;
  const bool hasB = i + d + 1 < n;
  const rnamc_fold_score_sets& f = b.params->contra;
  const float ebp = f.external_score_basepair, eun = f.external_score_unpair;
  const float mbp = f.multibranch_score_basepair, mun = f.multibranch_score_unpair;
  const float* __restrict__ qa = q.m[M_QA];
  float eA = kNegInf, mA = kNegInf, eB = kNegInf, mB = kNegInf;
  // k = i + t; j - k = d - t for the first cell, d + 1 - t for the second
  auto step = [&](float x, uint32_t t) {
    const float ca = static_cast<float>(d - t), cb = static_cast<float>(d + 1 - t);
    eA = lse(eA, x + ebp + eun * ca, tab);
    eB = lse(eB, x + ebp + eun * cb, tab);
    mA = lse(mA, x + mbp + mun * ca, tab);
    mB = lse(mB, x + mbp + mun * cb, tab);
  };
  struct ABuf {
    float xs[kU];
  };
  uint32_t t = pingpong<ABuf, kU, true>(
      1u, d / kU,
      [&](ABuf& B, uint32_t t0) {
#pragma unroll
        for (int u = 0; u < kU; u++) B.xs[u] = ldu(qa + tri_off(n, t0 + u), i * 4u);
      },
      [&](const ABuf& B, uint32_t t0) {
#pragma unroll
        for (int u = 0; u < kU; u++) step(B.xs[u], t0 + u);
      });
  for (; t <= d; t++) step(qa[tri_off(n, t) + i], t);
  const uint32_t odA = tri_off(n, d) + i, odB = tri_off(n, d + 1) + i;
  q.m[M_ZRE][odA] = eA;
  q.m[M_ZRM][odA] = mA;
  if (!hasB) return;
  {  // k = j + 1 of the second cell
    const float x = qa[odB];
    eB = lse(eB, x + ebp + eun * 0.f, tab);
    mB = lse(mB, x + mbp + mun * 0.f, tab);
  }
  q.m[M_ZRE][odB] = eB;
  q.m[M_ZRM][odB] = mB;
}

template <bool CONTRA>
__device__ __forceinline__ void inside_sums_pair2(const DeviceBatch& b, const Seq& q, uint32_t d,
                                                  uint32_t i, const LseTab* tab) {
  const uint32_t n = q.n;
  const uint32_t odA = tri_off(n, d) + i, odB = tri_off(n, d + 1) + i;
  const bool hasB = i + d + 1 < n;
  const float* __restrict__ zre = q.m[M_ZRE];
  const float* __restrict__ zrm = q.m[CONTRA ? M_ZRM : M_ZRE];
  const float* __restrict__ z = q.m[M_Z];
  const float* __restrict__ q1 = q.m[M_Q1D];
  const float* __restrict__ qa = q.m[M_QA];
  const float c = CONTRA ? 0.f : b.params->turner.coeff_num_branches;
  const float mun = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;

  float zeA, zmA, zeB, zmB, zrN = 0.f;
  if (!CONTRA) {
    zeA = lse(zre[tri_off(n, d - 1) + i], qa[odA], tab);
    zeB = lse(zeA, qa[odB], tab);  // pad makes the read safe when !hasB
    // Zr(i+1, j+1): diagonal d, offset i+1
    zrN = lse(zre[tri_off(n, d - 1) + i + 1], qa[odA + 1], tab);
    zmA = zeA;
    zmB = zeB;
    q.m[M_ZRE][odA] = zeA;
    if (hasB) q.m[M_ZRE][odB] = zeB;
  } else {
    zeA = zre[odA];
    zmA = zrm[odA];
    zeB = zre[odB];
    zmB = zrm[odB];
  }

  float extA, extB, s1A, s1B, s2A = kNegInf, s2B = kNegInf;
  if (!CONTRA) {
    extA = lse(0.f, zeA + 0.f, tab);  // k = i: Z[i][i-1] is the lower-triangle 0
    extB = lse(0.f, zeB + 0.f, tab);
    s1A = zeA + c;
    s1B = zeB + c;
  } else {
    const float eun = b.params->contra.external_score_unpair;
    extA = lse(eun * static_cast<float>(d + 1), zeA + 0.f, tab);
    extB = lse(eun * static_cast<float>(d + 2), zeB + 0.f, tab);
    s1A = zmA;
    s1B = zmB;
  }
  // ea/ma: Zr_ext / Zr_mb of column j, eb/mb of column j+1 (Turner: ma = ea, mb = eb)
  auto step = [&](float ea, float ma, float eb, float mb, float zz, float qq, uint32_t t) {
    extA = lse(extA, ea + zz, tab);
    extB = lse(extB, eb + zz, tab);
    if (!CONTRA) {
      const float xa = ea + c, xb = eb + c;
      s1A = lse(s1A, xa, tab);
      s1B = lse(s1B, xb, tab);
      s2A = lse(s2A, qq + xa, tab);
      s2B = lse(s2B, qq + xb, tab);
    } else {
      const float tt = mun * static_cast<float>(t);
      s1A = lse(s1A, ma + tt, tab);
      s1B = lse(s1B, mb + tt, tab);
      s2A = lse(s2A, qq + ma, tab);
      s2B = lse(s2B, qq + mb, tab);
    }
  };
  const uint32_t i4 = i * 4u;
  struct SBuf {
    float ea[kU], ma[kU], eb[kU], mb[kU], zz[kU], qq[kU];
  };
  // steps 1 .. d-1 of both cells: k = i + t
  uint32_t t = pingpong<SBuf, kU, true>(
      1u, (d - 1) / kU,
      [&](SBuf& B, uint32_t t0) {
#pragma unroll
        for (int u = 0; u < kU; u++) {
          const uint32_t oa = tri_off(n, d - t0 - u) + t0 + u;
          const uint32_t ob = tri_off(n, d + 1 - t0 - u) + t0 + u;
          const uint32_t o = tri_off(n, t0 + u - 1);
          B.ea[u] = ldu(zre + oa, i4);
          B.eb[u] = ldu(zre + ob, i4);
          B.ma[u] = CONTRA ? ldu(zrm + oa, i4) : 0.f;
          B.mb[u] = CONTRA ? ldu(zrm + ob, i4) : 0.f;
          B.zz[u] = ldu(z + o, i4);
          B.qq[u] = ldu(q1 + o, i4);
        }
      },
      [&](const SBuf& B, uint32_t t0) {
#pragma unroll
        for (int u = 0; u < kU; u++) {
          const float eb = (!CONTRA && t0 + u == 1u) ? zrN : B.eb[u];
          step(B.ea[u], B.ma[u], eb, B.mb[u], B.zz[u], B.qq[u], t0 + u);
        }
      });
  for (; t < d; t++) {
    const uint32_t oa = tri_off(n, d - t) + t + i, ob = tri_off(n, d + 1 - t) + t + i;
    const uint32_t o = tri_off(n, t - 1) + i;
    const float eb = (!CONTRA && t == 1u) ? zrN : zre[ob];
    step(zre[oa], CONTRA ? zrm[oa] : 0.f, eb, CONTRA ? zrm[ob] : 0.f, z[o], q1[o], t);
  }
  q.m[M_Z][odA] = extA;
  q.m[M_QM][odA] = s2A;
  s1A = lse(s1A, s2A, tab);
  q.m[M_Q1D][odA] = s1A;
  if (i >= 1) q.m[M_Q1C][col_off(i + d) + i - 1] = s1A;
  if (!hasB) return;
  {  // step t = d of the second cell: k = j
    const uint32_t ob = tri_off(n, 1) + d + i, o = tri_off(n, d - 1) + i;
    const float eb = (!CONTRA && d == 1u) ? zrN : zre[ob];
    const float mb = CONTRA ? zrm[ob] : 0.f;
    const float zz = z[o], qq = q1[o];
    extB = lse(extB, eb + zz, tab);
    if (!CONTRA) {
      const float xb = eb + c;
      s1B = lse(s1B, xb, tab);
      s2B = lse(s2B, qq + xb, tab);
    } else {
      s1B = lse(s1B, mb + mun * static_cast<float>(d), tab);
      s2B = lse(s2B, qq + mb, tab);
    }
  }
  q.m[M_Z][odB] = extB;
  q.m[M_QM][odB] = s2B;
  s1B = lse(s1B, s2B, tab);
  q.m[M_Q1D][odB] = s1B;
  if (i >= 1) q.m[M_Q1C][col_off(i + d + 1) + i - 1] = s1B;
}

// Latency form of the folds for launches too small to fill the chip (a single long
// sequence): the three chains of a cell run on three lanes (lanes 0-20 of a wave:
// sums_external, 21-41: sums_1ormore, 42-62: sums_multibranch, 21 cells per wave),
// so a k-step costs one dependent fold instead of three.  Same operations per chain
// as inside_sums_cell, bit for bit.
constexpr uint32_t kSplitCells = 21;
constexpr int kUL = 32;  // fold steps fetched ahead in the latency forms
template <bool CONTRA>
__device__ __forceinline__ void inside_sums_split(const DeviceBatch& b, const Seq& q, uint32_t d,
                                                  uint32_t cell0, const LseTab* tab) {
  const uint32_t n = q.n;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t chain = lane / kSplitCells;           // 0 ext, 1 s1, 2 s2, 3 idle
  const uint32_t i = cell0 + lane % kSplitCells;
  const bool valid = chain < 3 && i < n - d;
  const uint32_t ii = valid ? i : 0u;                  // idle lanes shadow cell 0 (no stores)
  const uint32_t od = tri_off(n, d) + ii;
  const float* __restrict__ zre = q.m[M_ZRE];
  const float* __restrict__ zrm = q.m[CONTRA ? M_ZRM : M_ZRE];
  const float* __restrict__ z = q.m[M_Z];
  const float* __restrict__ q1 = q.m[M_Q1D];
  const uint32_t i4 = ii * 4u;

  float zr = kNegInf;  // Turner: Zr_ext on every lane; CONTRAfold: Zr_ext on chain 0, Zr_mb on chain 1
  float c = 0.f, mun = 0.f;
  if (!CONTRA) {
    const float prev = (d >= 1) ? zre[tri_off(n, d - 1) + ii] : kNegInf;
    zr = lse(prev, q.m[M_QA][od], tab);
    if (valid && chain == 0) q.m[M_ZRE][od] = zr;
    c = b.params->turner.coeff_num_branches;
  } else {
    const rnamc_fold_score_sets& f = b.params->contra;
    mun = f.multibranch_score_unpair;
    // x + P + Q * cnt with (P, Q) = (ext_bp, ext_unpair) on chain 0, (mb_bp, mb_unpair) on chain 1
    const float P = (chain == 0) ? f.external_score_basepair : f.multibranch_score_basepair;
    const float Q = (chain == 0) ? f.external_score_unpair : mun;
    const bool live = valid && chain < 2;
    const float* __restrict__ qa = q.m[M_QA];
    struct ABuf {
      float xs[kUL];
    };
    uint32_t t = pingpong<ABuf, kUL>(
        1u, d / kUL,
        [&](ABuf& B, uint32_t t0) {
#pragma unroll
          for (int u = 0; u < kUL; u++) B.xs[u] = ldu(qa + tri_off(n, t0 + u), i4);
        },
        [&](const ABuf& B, uint32_t t0) {
#pragma unroll
          for (int u = 0; u < kUL; u++) {
            const float x = live ? B.xs[u] : kNegInf;
            zr = lse(zr, x + P + Q * static_cast<float>(d - t0 - u), tab);
          }
        });
    for (; t <= d; t++) {
      const float x = live ? qa[tri_off(n, t) + ii] : kNegInf;
      zr = lse(zr, x + P + Q * static_cast<float>(d - t), tab);
    }
    if (valid && chain == 0) q.m[M_ZRE][od] = zr;
    if (valid && chain == 1) q.m[M_ZRM][od] = zr;
  }

  // chain accumulators
  float acc;
  if (chain == 0) {
    acc = CONTRA ? lse(b.params->contra.external_score_unpair * static_cast<float>(d + 1), zr + 0.f, tab)
                 : lse(0.f, zr + 0.f, tab);
  } else if (chain == 1) {
    acc = CONTRA ? zr : zr + c;
  } else {
    acc = kNegInf;
  }
  // operands per lane: ra = Zr (ext flavour on chain 0, mb flavour otherwise under
  // CONTRAfold), rb = Z on chain 0, Q1 otherwise
  const float* __restrict__ pa = (CONTRA && chain != 0) ? zrm : zre;
  const float* __restrict__ pb = (chain == 0) ? z : q1;
  auto step = [&](float ra, float rb, uint32_t t) {
    const float x1 = CONTRA ? ra + mun * static_cast<float>(t) : ra + c;
    float term;
    if (!CONTRA) {
      term = (chain == 0) ? ra + rb : (chain == 1 ? x1 : rb + x1);
    } else {
      term = (chain == 1) ? x1 : rb + ra;
    }
    acc = lse(acc, term, tab);
  };
  // chunks of kUL steps: a lone wave per SIMD folds 8 steps in 0.7 us, less than one HBM
  // round trip, so the operands are fetched 32 steps (2.8 us) ahead
  struct SBuf {
    float ra[kUL], rb[kUL];
  };
  uint32_t t = pingpong<SBuf, kUL>(
      1u, d >= 1 ? (d - 1) / kUL : 0u,
      [&](SBuf& B, uint32_t t0) {
#pragma unroll
        for (int u = 0; u < kUL; u++) {
          B.ra[u] = pa[tri_off(n, d - t0 - u) + t0 + u + ii];
          B.rb[u] = pb[tri_off(n, t0 + u - 1) + ii];
        }
      },
      [&](const SBuf& B, uint32_t t0) {
#pragma unroll
        for (int u = 0; u < kUL; u++) step(B.ra[u], B.rb[u], t0 + u);
      });
  for (; t < d; t++) step(pa[tri_off(n, d - t) + t + ii], pb[tri_off(n, t - 1) + ii], t);
  // gather the three chains of a cell onto its chain-0 lane
  const float s1 = __shfl(acc, static_cast<int>((lane % kSplitCells) + kSplitCells));
  const float s2 = __shfl(acc, static_cast<int>((lane % kSplitCells) + 2 * kSplitCells));
  if (valid && chain == 0) {
    q.m[M_Z][od] = acc;
    q.m[M_QM][od] = s2;
    const float q1v = lse(s1, s2, tab);
    q.m[M_Q1D][od] = q1v;
    if (i >= 1) q.m[M_Q1C][col_off(i + d) + i - 1] = q1v;
  }
}

// One launch of the inside sweep: blocks [0, blocks_sums) fold diagonal d, the
// remaining blocks evaluate the closing-pair block of diagonal d+1, which needs
// nothing newer than diagonal d-1 (sums_multibranch[i+1][j-1], sums_close of
// spans <= d-1) and so runs beside the folds.
template <bool CONTRA, bool SPLIT>
__global__ void __launch_bounds__(256) k_inside(DeviceBatch b, uint32_t d, uint32_t blocks_sums,
                                                uint32_t nseq, int do_sums, int do_pair) {
  __shared__ LseTab tabs;
  __shared__ ProbeTabs L;
  const LseTab* tab = &tabs;
  load_lse_table(&tabs);
  // Blocks are dealt round-robin to the 8 XCDs by linear id.  Consecutive ids walk
  // over SEQUENCES (same role, same cell range), so every XCD receives the same mix
  // of light and heavy blocks whatever the grid width is.
  const uint32_t bxr = blockIdx.x / nseq;           // role + cell-range index
  const uint32_t which = blockIdx.x - bxr * nseq;   // sequence
  const Seq q = load_seq(b, which);
  const uint32_t n = q.n;
  if (bxr < blocks_sums) {
    if (!do_sums || d >= n) return;
    if (SPLIT) {
      const uint32_t wave = (bxr * blockDim.x + threadIdx.x) >> 6;
      const uint32_t cell0 = wave * kSplitCells;
      if (cell0 >= n - d) return;  // whole wave past the diagonal
      inside_sums_split<CONTRA>(b, q, d, cell0, tab);
    } else {
      const uint32_t i = bxr * blockDim.x + threadIdx.x;
      if (i >= n - d) return;
      inside_sums_cell<CONTRA>(b, q, d, i, tab);
    }
  } else {
    const uint32_t dp = d + 1;
    if (!do_pair || dp >= n) return;  // uniform over the block
    const uint32_t cnt = q.ccnt[dp];
    const uint32_t t = (bxr - blocks_sums) * blockDim.x + threadIdx.x;
    if ((bxr - blocks_sums) * blockDim.x >= cnt) return;  // block past the list
    load_probe_tabs<CONTRA, false>(L, b.params);
    if (t - (threadIdx.x & 63u) >= cnt) return;  // wave past the list
    uint32_t i;
    const bool valid = listed_cell(q, dp, t, cnt, i);
    inside_pair_cell<CONTRA, PAIR_FULL>(b, q, dp, i, valid, tab, L);
  }
}

// Two-diagonal launch: blocks [0, blocks_sums) fold diagonals d and d+1
// (inside_sums_pair2, or with ZR_ONLY the CONTRAfold rightmost-pair sums that precede
// them); then blocks_head blocks each for the early part of the closing-pair blocks of
// diagonals d+2 and d+3, which needs sums_close of spans <= d+1 only.
template <bool CONTRA, bool ZR_ONLY, bool HEADS>
__global__ void __launch_bounds__(256) k_inside2(DeviceBatch b, uint32_t d, uint32_t blocks_sums,
                                                 uint32_t blocks_head, uint32_t nseq, int do_sums,
                                                 int do_head) {
  __shared__ LseTab tabs;
  __shared__ ProbeTabStore<HEADS> Lstore;
  ProbeTabs& L = reinterpret_cast<ProbeTabs&>(Lstore);
  if (!HEADS) do_head = 0;
  load_lse_table(&tabs);
  uint32_t bxr = blockIdx.x / nseq;
  const uint32_t which = blockIdx.x - bxr * nseq;
  const Seq q = load_seq(b, which);
  const uint32_t n = q.n;
  if (do_head && !ZR_ONLY) {
    // dispatch order (ctx knob order_inside): 0 = folds, heads; 1 = heads, folds;
    // 2 = fold, head, head interleaved
    const uint32_t mode = static_cast<uint32_t>(b.order_inside);
    const uint32_t S = blocks_sums, H = 2u * blocks_head, p = bxr;
    if (mode == 1u) {
      bxr = (p < H) ? S + p : p - H;
    } else if (mode == 2u) {
      const uint32_t m = min(S, blocks_head);
      if (p < 3u * m) {
        const uint32_t r = p % 3u, x = p / 3u;
        bxr = (r == 0u) ? x : (r == 1u ? S + x : S + blocks_head + x);
      } else if (S > m) {
        bxr = m + (p - 3u * m);
      } else {
        const uint32_t q2 = p - 3u * m, hh = blocks_head - m;
        bxr = (q2 < hh) ? S + m + q2 : S + blocks_head + m + (q2 - hh);
      }
    }
  }
  if (bxr < blocks_sums) {
    const uint32_t i = bxr * blockDim.x + threadIdx.x;
    if (!do_sums || d >= n || i >= n - d) return;
    if (ZR_ONLY) {
      inside_zr_pair2(b, q, d, i, &tabs);
    } else {
      inside_sums_pair2<CONTRA>(b, q, d, i, &tabs);
    }
  } else {
    const uint32_t hb = bxr - blocks_sums;
    const uint32_t dp = d + 2 + hb / blocks_head;
    const uint32_t blk = hb % blocks_head;
    if (!HEADS || !do_head || dp >= n) return;
    const uint32_t cnt = q.ccnt[dp];
    if (blk * blockDim.x >= cnt) return;
    if (HEADS) {
      load_probe_tabs<CONTRA, false>(L, b.params);
      const uint32_t t = blk * blockDim.x + threadIdx.x;
      if (t - (threadIdx.x & 63u) >= cnt) return;
      uint32_t i;
      const bool valid = listed_cell(q, dp, t, cnt, i);
      inside_pair_cell<CONTRA, PAIR_HEAD>(b, q, dp, i, valid, &tabs, L);
    }
  }
}

// multibranch term of the closing-pair blocks of diagonals d0 .. d0+nd-1 whose early part
// is parked (two-diagonal schedule)
template <bool CONTRA>
__global__ void __launch_bounds__(256) k_pair_tail(DeviceBatch b, uint32_t d0, uint32_t blocks_d,
                                                   uint32_t nseq) {
  __shared__ LseTab tabs;
  __shared__ NoProbeTabs Lstore;  // PAIR_TAIL folds no probes: the cell function never reads it
  const ProbeTabs& L = reinterpret_cast<const ProbeTabs&>(Lstore);
  load_lse_table(&tabs);
  const uint32_t bxr = blockIdx.x / nseq;
  const uint32_t which = blockIdx.x - bxr * nseq;
  const Seq q = load_seq(b, which);
  const uint32_t dp = d0 + bxr / blocks_d;
  const uint32_t blk = bxr % blocks_d;
  if (dp >= q.n) return;
  const uint32_t cnt = q.ccnt[dp];
  const uint32_t t = blk * blockDim.x + threadIdx.x;
  if (t - (threadIdx.x & 63u) >= cnt) return;
  uint32_t i;
  const bool valid = listed_cell(q, dp, t, cnt, i);
  inside_pair_cell<CONTRA, PAIR_TAIL>(b, q, dp, i, valid, &tabs, L);
}

// ----------------------------------------------------------------------------
// outside pass (src/mccaskill_algo.rs:537-605 Turner, 638-718 CONTRAfold).
// basepair_probs stays in the log domain in `out` until k_finalize.

// probs_multibranch / probs_multibranch2 of one cell: k = j + t over pairs (i,k).
// Loads are unconditional (a lane past its own row end reads a neighbouring
// diagonal: in bounds thanks to the 64-float pad of every matrix) and the lane's
// validity is applied to the value, so the loop body has no exec-mask branches and
// the 2 x 8 loads of a chunk issue back to back.
template <bool CONTRA>
__device__ __forceinline__ void outside_mb_cell(const DeviceBatch& b, const Seq& q, uint32_t d,
                                                uint32_t i, bool valid, uint32_t cnt_wave,
                                                const LseTab* tab) {
  const uint32_t n = q.n;
  const uint32_t j = i + d;
  const float* __restrict__ q1d = q.m[M_Q1D];
  const float* __restrict__ w = q.m[M_W];
  const uint32_t* __restrict__ pk = q.pk;
  // W is stored for canonical cells only (62 % of a diagonal is no pair): cell (i,k) of
  // diagonal D = k-i sits at list index c64[i/64][D] + (canonical cells of this wave's
  // lanes before this one), which is a ballot away — the bases come from the 2-bit copy
  const uint32_t* __restrict__ c64row =
      q.c64 + static_cast<size_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(i >> 6))) * (n + 64u);
  const float mun = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  const uint32_t cnt = valid ? n - 1 - j : 0u;  // this lane's trip count
  const uint32_t i4 = i * 4u;
  // partners of base a as a 4-bit set: A:{U} C:{G} G:{C,U} U:{A,G}
  const uint32_t pairmask = valid ? (0x5A48u >> (4u * q.s[i])) & 15u : 0u;
  float pm = kNegInf, pm2 = kNegInf;
#ifdef RNAMC_COUNT_FAR
  FarCount fc;
#endif
  auto step = [&](float x, float r, uint32_t t) {
#ifdef RNAMC_COUNT_FAR
    fc.fold(pm, x + r);
    fc.fold(pm2, CONTRA ? x + mun * static_cast<float>(t - 1) : x);
    fc.end_k();
#endif
    pm = lse_wu(pm, x + r, tab);
    if (CONTRA) {
      pm2 = lse_wu(pm2, x + mun * static_cast<float>(t - 1), tab);
    } else {
      pm2 = lse_wu(pm2, x, tab);
    }
  };
  struct DAux {
    uint32_t t0;        // first step of the chunk
    uint32_t win;       // bases k = j+t0 .. j+t0+15, 2 bits each
    uint32_t base[kU];  // c64 of diagonals d+t0 .. d+t0+kU-1
  };
  struct DOps {
    float xs[kU], rs[kU];
    uint32_t bits;
  };
  auto load_aux = [&](DAux& A, uint32_t t0) {
    A.t0 = t0;
    const uint32_t bit = 2u * (j + t0 + 32u);
    const uint32_t wd = bit >> 5, sh = bit & 31u;
    A.win = __builtin_amdgcn_alignbit(pk[wd + 1], pk[wd], sh);
#pragma unroll
    for (int u = 0; u < kU; u++) A.base[u] = c64row[d + t0 + u];
  };
  auto locate = [&](uint32_t win, uint32_t u, uint32_t t, uint32_t base, bool& has) {
    const uint32_t c = (win >> (2u * u)) & 3u;
    has = (((pairmask >> c) & 1u) != 0u) && (t <= cnt);
    const unsigned long long m = __ballot(has);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi(
        static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
    return (base + rank) * 4u;
  };
  auto issue = [&](DOps& B, const DAux& A) {
    const uint32_t t0 = A.t0;
    uint32_t bits = 0, off[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      bool has;
      off[u] = locate(A.win, u, t0 + u, A.base[u], has);
      bits |= (has ? 1u : 0u) << u;
    }
    B.bits = bits;
    // lanes whose row ended before this chunk (the upper lanes of the wave: the walk runs
    // to the first lane's end) issue no loads; their buffers keep finite values and their
    // terms are masked by `bits`
    if (t0 <= cnt) {
#pragma unroll
      for (int u = 0; u < kU; u++) {
        B.xs[u] = ldu(w + tri_off(n, d + t0 + u), off[u]);
        B.rs[u] = ldu(q1d + tri_off(n, t0 + u - 2) + d + 1, i4);
      }
    }
  };
  auto fold = [&](const DOps& B, uint32_t t0) {
#pragma unroll
    for (int u = 0; u < kU; u++)
      step(((B.bits >> u) & 1u) ? B.xs[u] : kNegInf, B.rs[u], t0 + u);
#ifdef RNAMC_COUNT_FAR
    fc.end_chunk();
#endif
  };
  auto single = [&](uint32_t t, float& x) {  // one step outside the pipeline
    const uint32_t bit = 2u * (j + t + 32u);
    const uint32_t wd = bit >> 5, sh = bit & 31u;
    const uint32_t win = __builtin_amdgcn_alignbit(pk[wd + 1], pk[wd], sh);
    bool has;
    const uint32_t off = locate(win, 0u, t, c64row[d + t], has);
    const float v = ldu(w + tri_off(n, d + t), off);
    x = has ? v : kNegInf;
  };
  if (cnt_wave >= 1) {
    // t = 1: sums_1ormore_basepairs[j+1][j] is the empty interval (-inf): only pm2 moves
    float x1;
    single(1u, x1);
    if (CONTRA) {
      pm2 = lse(pm2, x1 + mun * 0.f, tab);
    } else {
      pm2 = lse(pm2, x1, tab);
    }
  }
  // steps 2 .. cnt_wave in chunks of kU, three stages deep: the window / c64 entries of
  // chunk c+2 and the operands of chunk c+1 are in flight while chunk c is folded.  Every
  // stage runs unconditionally (past the last chunk it re-fetches the last chunk), so the
  // number of loads in flight at each wait is static.
  uint32_t t = 2u;
  const uint32_t nch = cnt_wave >= 2 ? (cnt_wave - 1) / kU : 0u;
  if (nch) {
    const uint32_t t_last = 2u + (nch - 1u) * kU;
    DAux xa, xb;
    DOps oa, ob;
#pragma unroll
    for (int u = 0; u < kU; u++) oa.xs[u] = oa.rs[u] = ob.xs[u] = ob.rs[u] = 0.f;
    load_aux(xa, t);
    load_aux(xb, min(t + kU, t_last));
    issue(oa, xa);
    uint32_t c = 0;
    for (;;) {
      load_aux(xa, min(t + 2 * kU, t_last));
      issue(ob, xb);
      fold(oa, t);
      t += kU;
      if (++c >= nch) break;
      load_aux(xb, min(t + 2 * kU, t_last));
      issue(oa, xa);
      fold(ob, t);
      t += kU;
      if (++c >= nch) break;
    }
  }
  for (; t <= cnt_wave; t++) {
    float x;
    single(t, x);
    step(x, ldu(q1d + tri_off(n, t - 2) + d + 1, i4), t);
  }
#ifdef RNAMC_COUNT_FAR
  fc.flush(1);
#endif
  if (valid) {
    // {probs_multibranch, probs_multibranch2} interleaved: 16 k-steps of a column = one
    // 128-byte line (slots PM and PM2 are adjacent and form one float2 array)
    reinterpret_cast<float2*>(q.m[M_PM])[col_off(j) + i] = make_float2(pm, pm2);
  }
}

// pair probability of one cell, first half: exterior term ⊕ enclosing 2-loops
// (559-593 / 663-700).  Needs only results of spans >= span+2, so it runs one
// launch ahead of the second half and parks the running sum in the log-prob slot.
template <bool CONTRA>
__device__ __forceinline__ void outside_pair_head(const DeviceBatch& b, const Seq& q, uint32_t d,
                                                  uint32_t i, bool valid, const LseTab* tab,
                                                  const ProbeTabs& L) {
  const uint32_t n = q.n;
  const uint32_t j = i + d;
  const uint32_t od = tri_off(n, d) + i;
  const float qb_ij = valid ? q.m[M_QB][od] : kNegInf;
  const bool paired = qb_ij > kNegInf;
  if (__ballot(paired) == 0ull) return;  // wave-uniform
  float p = kNegInf;
  if (paired) {
    const float qa_ij = q.m[M_QA][od];
    const float* z = q.m[M_Z];
    const float ztot = z[tri_off(n, n - 1)];
    const float zl = (i < 1) ? 0.f : z[tri_off(n, i - 1)];                  // Z[0][i-1]
    const float zr = (j > n - 2) ? 0.f : z[tri_off(n, n - 2 - j) + j + 1];  // Z[j+1][n-1]
    if (CONTRA) {
      p = zl + zr + qa_ij + b.params->contra.external_score_basepair - ztot;
    } else {
      p = zl + qa_ij + zr - ztot;
    }
  }
  // enclosing pairs (k,l) = (i-1-a, j+1+bb), a ascending (k descending), bb ascending,
  // a + bb <= 30, k >= 0, l <= n-1; rows with d+2+a > n-1 have no diagonal left
  if (d + 2 < n) {
    const uint32_t lim = min(static_cast<uint32_t>(RNAMC_MAX_2LOOP_LEN), n - 3 - d);
    p = probe_fold<CONTRA, true>(b, q, d, i, paired, lim, p, qb_ij, tab, L);
  }
  if (paired) q.m[M_P][od] = p;
}

// pair probability of one cell, second half: the multibranch contexts k = 0..i-1
// (594-605 / 701-718), continuing the fold parked by outside_pair_head:
//   x  = sums_1ormore_basepairs[k+1][i-1]  row-major row k+1 (empty when k+1 > i-1)
//   y2 = probs_multibranch2[k][j], y = probs_multibranch[k][j]   row-major row k
// A lane that is no pair, or is past its own i, folds -inf terms (no-ops).
template <bool CONTRA>
__device__ __forceinline__ void outside_pair_tail(const DeviceBatch& b, const Seq& q, uint32_t d,
                                                  uint32_t i, uint32_t tl, bool valid,
                                                  uint32_t imax_wave, const LseTab* tab) {
  const uint32_t n = q.n;
  const uint32_t od = tri_off(n, d) + i;
  const float qb_ij = valid ? q.m[M_QB][od] : kNegInf;
  const bool paired = qb_ij > kNegInf;
  if (__ballot(paired) == 0ull) return;  // wave-uniform
  float p = kNegInf, sa = kNegInf, mun = 0.f;
  if (paired) {
    p = q.m[M_P][od];
    const float qa_ij = q.m[M_QA][od];
    if (CONTRA) {
      sa = qa_ij + b.params->contra.multibranch_score_basepair;
    } else {
      sa = qa_ij + b.params->turner.coeff_num_branches;
    }
  }
  if (CONTRA) mun = b.params->contra.multibranch_score_unpair;
  // Column walks: probs_multibranch{,2}[k][j] = column j, sums_1ormore_basepairs[k+1][i-1]
  // = column i-1 stored one row up.  Each lane streams its own three columns in
  // 32-byte pieces (two dwordx4 per column and chunk); a column's padding makes a read
  // past the lane's own rows harmless, validity is applied to the values:
  //   k < iend   : the step exists for this lane
  //   k+2 <= iend: the interval [k+1, i-1] is not empty
  const uint32_t iend = paired ? i : 0u;
  const uint32_t j = i + d;
  // {y, y2} pairs of column j: float4 = two k-steps
  const float4* __restrict__ yycol =
      reinterpret_cast<const float4*>(reinterpret_cast<const float2*>(q.m[M_PM]) + col_off(j));
  const float4* __restrict__ xcol =
      reinterpret_cast<const float4*>(q.m[M_Q1C] + col_off(i >= 1 ? i - 1 : 0));
#ifdef RNAMC_COUNT_FAR
  FarCount fc;
#endif
  auto step = [&](float x, float y, float y2, uint32_t k) {
    const bool vy = k < iend;
    x = (k + 2 <= iend) ? x : kNegInf;
    y = vy ? y : kNegInf;
    y2 = vy ? y2 : kNegInf;
#ifdef RNAMC_COUNT_FAR
    fc.fold(p, sa + y2 + x);
#endif
    p = lse_wu(p, sa + y2 + x, tab);
    if (CONTRA) {
#ifdef RNAMC_COUNT_FAR
      fc.fold(p, sa + y + mun * static_cast<float>(i - k - 1));
#endif
      p = lse_wu(p, sa + y + mun * static_cast<float>(i - k - 1), tab);
    } else {
#ifdef RNAMC_COUNT_FAR
      fc.fold(p, sa + y);
#endif
      p = lse_wu(p, sa + y, tab);
    }
#ifdef RNAMC_COUNT_FAR
    fc.fold(p, sa + x + y);
    fc.end_k();
#endif
    p = lse_wu(p, sa + x + y, tab);
  };
  // Chunks of 16 k-steps.  A 128-byte line holds 16 steps of {y, y2} and 32 steps of x; a
  // lane fetches whole lines of its own columns (64-byte pieces read 11 % slower) into ONE
  // buffer per stream (64 VGPRs).  Double-buffering them (112 VGPRs, 3 waves per SIMD) was
  // 12 % slower over the whole outside sweep: the other waves of the SIMD hide the fetch
  // latency better than a deeper pipeline that starves every role of registers (and on a
  // single long sequence, where this chain is the critical path, it gained nothing).
  const uint32_t nch = (imax_wave + 15u) / 16u;
  if (nch) {
    float4 xl[8], yl[8];
    auto fold16l = [&](const float4* x, const float4(&y)[8], uint32_t k0) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        step(x[u].x, y[2 * u].x, y[2 * u].y, k0 + 4 * u);
        step(x[u].y, y[2 * u].z, y[2 * u].w, k0 + 4 * u + 1);
        step(x[u].z, y[2 * u + 1].x, y[2 * u + 1].y, k0 + 4 * u + 2);
        step(x[u].w, y[2 * u + 1].z, y[2 * u + 1].w, k0 + 4 * u + 3);
      }
    };
    for (uint32_t c = 0, k = 0; c < nch; c++, k += 16) {
      // the streams are lane-private, so a lane past its own walk (the wave runs to its
      // longest lane's end) issues no loads: nothing is fetched for steps that fold nothing
      if (k < iend) {
        if ((c & 1u) == 0u) {
#pragma unroll
          for (int u = 0; u < 8; u++) xl[u] = xcol[k / 4 + u];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) yl[u] = yycol[k / 2 + u];
      }
      if ((c & 1u) == 0u) fold16l(&xl[0], yl, k); else fold16l(&xl[4], yl, k);
#ifdef RNAMC_COUNT_FAR
      fc.end_chunk();
#endif
    }
  }
#ifdef RNAMC_COUNT_FAR
  fc.flush(0);
#endif
  if (paired && p > kNegInf) {
    q.m[M_P][od] = p;
    reinterpret_cast<float2*>(q.m[M_PQ])[od] = make_float2(p, qb_ij);
    q.m[M_W][tri_off(n, d) + tl] = p + q.m[M_MBC][od] - qb_ij;  // list index, see outside_mb_cell
  }
}

// One launch of the outside sweep: three independent roles in disjoint blocks —
// probs_multibranch{,2} of diagonal d, the multibranch half of the pair
// probabilities of diagonal d, and the 2-loop half of diagonal d-1 (which the
// next launch continues).  All read only results of longer spans.
// ROLES (compile time): bit 0 probs_multibranch, bit 1 pair tail, bit 2 pair head.  The pair
// tail keeps whole cache lines of three columns in registers; compiled into one kernel it
// sets the occupancy of the other two roles as well, so large launches run it as a kernel
// of its own on a second stream (ROLES = 2 beside ROLES = 5) and only small launches use
// the all-in-one form (ROLES = 7).
template <bool CONTRA, int ROLES>
__global__ void __launch_bounds__(256)
    __attribute__((amdgpu_waves_per_eu((ROLES == 5 || ROLES == 1) ? 4 : 1, 8))) k_outside(DeviceBatch b, uint32_t d, uint32_t blocks_mb,
                                                 uint32_t blocks_head, uint32_t nseq, int do_mb,
                                                 int do_tail, int do_head) {
  __shared__ LseTab tabs;
  __shared__ ProbeTabStore<(ROLES & 4) != 0> Lstore;
  ProbeTabs& L = reinterpret_cast<ProbeTabs&>(Lstore);
  const LseTab* tab = &tabs;
  load_lse_table(&tabs);
  // linear id -> (role block, sequence) as in k_inside; within a role the blocks
  // with the longest walks are issued first (probs_multibranch: small i; pair
  // probabilities: large i)
  uint32_t bxr = blockIdx.x / nseq;
  const uint32_t which = blockIdx.x - bxr * nseq;
  const Seq q = load_seq(b, which);
  const uint32_t n = q.n;
  {
    // dispatch order (ctx knob order_outside): 0 = D, E, head; 1 = E, head, D;
    // 2 = E and D interleaved, then head; 3 = head, then E/D interleaved; 4 = three-way
    const uint32_t mode = static_cast<uint32_t>(b.order_outside);
    const uint32_t B = blocks_mb, Bh = blocks_head;
    const uint32_t p = bxr;
    if (mode == 1u) {
      if (p < B) bxr = B + p;
      else if (p < B + Bh) bxr = 2u * B + (p - B);
      else if (p < 2u * B + Bh) bxr = p - B - Bh;
    } else if (mode == 2u) {
      if (p < 2u * B) bxr = (p & 1u) ? B + (p >> 1) : (p >> 1);
    } else if (mode == 3u) {
      if (p < Bh) bxr = 2u * B + p;
      else if (p < 2u * B + Bh) {
        const uint32_t q2 = p - Bh;
        bxr = (q2 & 1u) ? B + (q2 >> 1) : (q2 >> 1);
      }
    } else if (mode == 4u) {
      const uint32_t m = min(B, Bh);
      if (p < 3u * m) {
        const uint32_t r = p % 3u, x = p / 3u;
        bxr = (r == 0u) ? B + x : (r == 1u ? x : 2u * B + x);
      } else if (B > Bh) {
        const uint32_t q2 = p - 3u * m;
        if (q2 < 2u * (B - m)) bxr = (q2 & 1u) ? B + m + (q2 >> 1) : m + (q2 >> 1);
      } else {
        const uint32_t q2 = p - 3u * m;
        if (q2 < Bh - m) bxr = 2u * B + m + q2;
      }
    }
  }
  if (bxr < blocks_mb) {
    if (!(ROLES & 1) || !do_mb || d >= n) return;
    const uint32_t cells = n - d;
    const uint32_t i = bxr * blockDim.x + threadIdx.x;
    const uint32_t wave_first = i - (threadIdx.x & 63u);  // lane 0 of this wave
    if (wave_first >= cells) return;                       // whole wave has no cell
    // the first lane of the wave has the longest walk: n-1-j with j = i+d
    // (readfirstlane: the value is wave-uniform, but derived from threadIdx the compiler
    // would keep the loop control and the diagonal offsets of the walk in vector registers)
    const uint32_t cnt_wave = static_cast<uint32_t>(
        __builtin_amdgcn_readfirstlane(static_cast<int>(n - 1 - d - wave_first)));
    outside_mb_cell<CONTRA>(b, q, d, i, i < cells, cnt_wave, tab);
  } else if (bxr < 2u * blocks_mb) {
    if (!(ROLES & 2) || !do_tail || d >= n) return;
    const uint32_t cnt = q.ccnt[d];
    const uint32_t bx = 2u * blocks_mb - 1u - bxr;  // descending: heavy blocks first
    if (bx * blockDim.x >= cnt) return;            // block past the list (uniform)
    const uint32_t t = bx * blockDim.x + threadIdx.x;
    const uint32_t wave_first = t - (threadIdx.x & 63u);
    if (wave_first >= cnt) return;
    uint32_t i;
    const bool valid = listed_cell(q, d, t, cnt, i);
    // the last listed lane of the wave has the largest i, i.e. the longest walk
    const uint32_t last_lane = min(63u, cnt - 1u - wave_first);
    const uint32_t imax = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
        __shfl(static_cast<int>(i), static_cast<int>(last_lane))));
    outside_pair_tail<CONTRA>(b, q, d, i, t, valid, imax, tab);
  } else {
    if (!(ROLES & 4) || !do_head || d == 0 || d - 1 >= n) return;
    const uint32_t dh = d - 1;
    const uint32_t cnt = q.ccnt[dh];
    const uint32_t bx = bxr - 2u * blocks_mb;
    if (bx >= blocks_head || bx * blockDim.x >= cnt) return;  // block past the list (uniform)
    if (ROLES & 4) {
      load_probe_tabs<CONTRA, true>(L, b.params);
      const uint32_t t = bx * blockDim.x + threadIdx.x;
      if (t - (threadIdx.x & 63u) >= cnt) return;
      uint32_t i;
      const bool valid = listed_cell(q, dh, t, cnt, i);
      outside_pair_head<CONTRA>(b, q, dh, i, valid, tab, L);
    }
  }
}


#include "rnamc_latency.h"

// Latency-form launches, one wave per workgroup (rnamc_latency.h).
// Inside, diagonal d: blocks [0, 3 cells) one (cell, role) chain each; then one lane per cell of
// diagonal d-1 for the combine that completes sums_1ormore_basepairs (do_combine).
// With pair_cells != 0 the launch opens with one block per cell of diagonal pair_d for its
// closing-pair blocks (what k_pair_lat would run beside this launch on a second stream: they
// need nothing newer than diagonal d-1 either).
template <bool CONTRA>
__global__ void __launch_bounds__(64) k_inside_lat(DeviceBatch b, uint32_t d, uint32_t cells_max,
                                                   uint32_t nseq, int form, int do_combine,
                                                   uint32_t pair_d, uint32_t pair_cells) {
  __shared__ LseTab tabs;
  uint32_t bx = blockIdx.x / nseq;
  const uint32_t which = blockIdx.x - bx * nseq;
  const Seq q = load_seq(b, which);
  const uint32_t n = q.n;
  if (bx < pair_cells) {
    if (pair_d >= n) return;
    const uint32_t cnt = q.ccnt[pair_d];
    if (bx >= cnt) return;
    const uint32_t i = uni(static_cast<uint32_t>(q.cidx[tri_off(n, pair_d) + bx]));
    inside_pair_lat<CONTRA, PAIR_FULL>(b, q, pair_d, i, load_piece8());
    return;
  }
  bx -= pair_cells;
  // form 2: eight cells of one role per wave; 0: no chains; form 2 | 4 (CONTRAfold): roles 3, 4
  // as well (the sums_rightmost_basepairs folds of diagonal d + 1 up to their last step); | 8:
  // this diagonal's were parked by the launch before
  const bool ahead = CONTRA && (form & 7) == 6, parked = CONTRA && (form & 11) == 10;
  form &= 3;
  const uint32_t per = 8u;
  const uint32_t kinds = ahead ? 5u : 3u;
  const uint32_t wpr = (cells_max + per - 1u) / per;  // waves per role
  const uint32_t nchain = form ? kinds * wpr : 0u;
  if (bx < nchain) {
    const uint32_t w = bx / kinds, role = bx - kinds * w;
    if (d >= n || w * per >= n - d) return;
    const Piece8 P8 = load_piece8();
    if (role == 0) inside_chain_e<CONTRA, 0>(b, q, d, w * 8u, P8, parked);
    else if (role == 1) inside_chain_e<CONTRA, 1>(b, q, d, w * 8u, P8, parked);
    else if (role == 2) inside_chain_e<CONTRA, 2>(b, q, d, w * 8u, P8, parked);
    else if (CONTRA && role == 3) inside_chain_e<CONTRA, 3>(b, q, d, w * 8u, P8, false);
    else if (CONTRA) inside_chain_e<CONTRA, 4>(b, q, d, w * 8u, P8, false);
  } else {
    if (!do_combine || d == 0 || d - 1 >= n) return;
    const uint32_t i0 = (bx - nchain) * 64u;
    if (i0 >= n - (d - 1)) return;
    load_lse_table(&tabs);
    const uint32_t i = i0 + (threadIdx.x & 63u);
    if (i < n - (d - 1)) inside_combine_lat(q, d - 1, i, &tabs);
  }
}

// Outside, diagonal d: blocks [0, cells): multibranch half of the pair probability of one
// listed cell each, longest chains (largest i) first; then one block per cell for
// probs_multibranch{,2}.
// With head_cells != 0 the launch also holds one block per cell of diagonal head_d (= d - 1)
// for the 2-loop half of its pair probabilities (k_pair_lat<., true>'s work: it needs both
// halves of diagonal d + 1 and beyond, like everything else in this launch).
template <bool CONTRA>
__global__ void __launch_bounds__(64) k_outside_lat(DeviceBatch b, uint32_t d, uint32_t cells_max,
                                                    uint32_t nseq, int do_mb, int do_tail,
                                                    uint32_t head_d, uint32_t head_cells) {
  uint32_t bx = blockIdx.x / nseq;
  const uint32_t which = blockIdx.x - bx * nseq;
  const Seq q = load_seq(b, which);
  const uint32_t n = q.n;
  const Piece8 P8 = load_piece8();
  if (bx >= 2u * cells_max) {  // after the chains of diagonal d
    bx -= 2u * cells_max;
    if (bx >= head_cells || head_d >= n) return;
    const uint32_t cnt = q.ccnt[head_d];
    if (bx >= cnt) return;
    const uint32_t i = uni(static_cast<uint32_t>(q.cidx[tri_off(n, head_d) + bx]));
    outside_head_lat<CONTRA>(b, q, head_d, i, P8);
    return;
  }
  if (d >= n) return;
  if (bx < cells_max) {
    if (!do_tail) return;
    const uint32_t cnt = q.ccnt[d];
    if (bx >= cnt) return;
    const uint32_t i = uni(static_cast<uint32_t>(q.cidx[tri_off(n, d) + (cnt - 1u - bx)]));
    // the launch's critical path: this wave issues whenever it can, the probs_multibranch
    // waves that share its SIMD take the gaps
    __builtin_amdgcn_s_setprio(3);
    outside_tail_lat<CONTRA>(b, q, d, i, P8);
  } else {
    if (!do_mb) return;
    const uint32_t i = bx - cells_max;
    if (i >= n - d) return;
    outside_mb_lat<CONTRA>(b, q, d, i, P8);
  }
}

// ----------------------------------------------------------------------------
// Durbin pair-HMM (src/durbin_algo.rs:79-242): forward and backward sums of one sequence
// pair by anti-diagonal (cell (i,j) needs (i-1,j-1), (i-1,j), (i,j-1) resp. the +1
// neighbours), one workgroup per (pair, direction), lanes = cells of the diagonal; then the
// match probabilities, one lane per cell.  Six row-major n1 x n2 matrices per pair in the
// workspace.  Every logsumexp fold has the reference's terms in the reference's order.
constexpr uint32_t kDurbinThreads = 1024;

__global__ void __launch_bounds__(kDurbinThreads)
    k_durbin_sums(const DurbinPair* __restrict__ pairs, const uint8_t* __restrict__ bases,
                  float* __restrict__ ws, rnamc_align_scores sc) {
  __shared__ LseTab tabs;
  load_lse_table(&tabs);
  const LseTab* tab = &tabs;
  const DurbinPair pr = pairs[blockIdx.x >> 1];
  const bool backward = (blockIdx.x & 1u) != 0u;
  const uint32_t n1 = pr.n1, n2 = pr.n2;
  const uint8_t* __restrict__ a = bases + pr.a_off;
  const uint8_t* __restrict__ b = bases + pr.b_off;
  const size_t cells = static_cast<size_t>(n1) * n2;
  float* __restrict__ m = ws + pr.ws_off + (backward ? 3 * cells : 0);
  float* __restrict__ mi = m + cells;
  float* __restrict__ md = m + 2 * cells;
  // AlignSums::new (60-71)
  for (size_t x = threadIdx.x; x < 3 * cells; x += blockDim.x) m[x] = kNegInf;
  __syncthreads();
  auto at = [n2](float* p, uint32_t i, uint32_t j) -> float& { return p[static_cast<size_t>(i) * n2 + j]; };
  if (!backward) {
    // cells 0 <= i <= n1-2, 0 <= j <= n2-2 (92-93), diagonal s = i + j
    for (uint32_t s = 0; s + 4 <= n1 + n2; s++) {
      const uint32_t ilo = s > n2 - 2 ? s - (n2 - 2) : 0u, ihi = min(s, n1 - 2);
      for (uint32_t i = ilo + threadIdx.x; i <= ihi; i += blockDim.x) {
        const uint32_t j = s - i;
        if (i == 0 && j == 0) {
          at(m, 0, 0) = 0.f;
          continue;
        }
        if (i > 0 && j > 0) {
          float sum = kNegInf;
          const bool begins = (i - 1 == 0) && (j - 1 == 0);
          sum = lse(sum, at(m, i - 1, j - 1) + (begins ? sc.init_match_score : sc.match2match_score), tab);
          sum = lse(sum, at(mi, i - 1, j - 1) + sc.match2insert_score, tab);
          sum = lse(sum, at(md, i - 1, j - 1) + sc.match2insert_score, tab);
          at(m, i, j) = sum + sc.match_scores[a[i]][b[j]];
        }
        if (i > 0) {
          float sum = kNegInf;
          const bool begins = (i - 1 == 0) && (j == 0);
          sum = lse(sum, at(m, i - 1, j) + (begins ? sc.init_insert_score : sc.match2insert_score), tab);
          sum = lse(sum, at(mi, i - 1, j) + sc.insert_extend_score, tab);
          at(mi, i, j) = sum + sc.insert_scores[a[i]];
        }
        if (j > 0) {
          float sum = kNegInf;
          const bool begins = (i == 0) && (j - 1 == 0);
          sum = lse(sum, at(m, i, j - 1) + (begins ? sc.init_insert_score : sc.match2insert_score), tab);
          sum = lse(sum, at(md, i, j - 1) + sc.insert_extend_score, tab);
          at(md, i, j) = sum + sc.insert_scores[b[j]];
        }
      }
      __syncthreads();
    }
  } else {
    // cells 1 <= i <= n1-1, 1 <= j <= n2-1 (152-153), diagonal s = i + j descending
    for (uint32_t s = n1 + n2 - 2; s >= 2; s--) {
      const uint32_t ilo = s > n2 - 1 ? s - (n2 - 1) : 1u, ihi = min(s - 1, n1 - 1);
      for (uint32_t i = ilo + threadIdx.x; i <= ihi; i += blockDim.x) {
        const uint32_t j = s - i;
        if (i == n1 - 1 && j == n2 - 1) {
          at(m, i, j) = 0.f;
          continue;
        }
        if (i < n1 - 1 && j < n2 - 1) {
          float sum = kNegInf;
          const bool ends = (i + 1 == n1 - 1) && (j + 1 == n2 - 1);
          sum = lse(sum, at(m, i + 1, j + 1) + (ends ? 0.f : sc.match2match_score), tab);
          sum = lse(sum, at(mi, i + 1, j + 1) + sc.match2insert_score, tab);
          sum = lse(sum, at(md, i + 1, j + 1) + sc.match2insert_score, tab);
          at(m, i, j) = sum + sc.match_scores[a[i]][b[j]];
        }
        if (i < n1 - 1) {
          float sum = kNegInf;
          const bool ends = (i + 1 == n1 - 1) && (j == n2 - 1);
          sum = lse(sum, at(m, i + 1, j) + (ends ? 0.f : sc.match2insert_score), tab);
          sum = lse(sum, at(mi, i + 1, j) + sc.insert_extend_score, tab);
          at(mi, i, j) = sum + sc.insert_scores[a[i]];
        }
        if (j < n2 - 1) {
          float sum = kNegInf;
          const bool ends = (i == n1 - 1) && (j + 1 == n2 - 1);
          sum = lse(sum, at(m, i, j + 1) + (ends ? 0.f : sc.match2insert_score), tab);
          sum = lse(sum, at(md, i, j + 1) + sc.insert_extend_score, tab);
          at(md, i, j) = sum + sc.insert_scores[b[j]];
        }
      }
      __syncthreads();
    }
  }
}

// get_match_probs (src/durbin_algo.rs:201-242)
__global__ void __launch_bounds__(256)
    k_durbin_probs(const DurbinPair* __restrict__ pairs, const float* __restrict__ ws,
                   float* __restrict__ out, rnamc_align_scores sc) {
  __shared__ LseTab tabs;
  load_lse_table(&tabs);
  const LseTab* tab = &tabs;
  const DurbinPair pr = pairs[blockIdx.y];
  const uint32_t n1 = pr.n1, n2 = pr.n2;
  const size_t cells = static_cast<size_t>(n1) * n2;
  const float* fm = ws + pr.ws_off;
  const float* fi = fm + cells;
  const float* fd = fm + 2 * cells;
  const float* bm = fm + 3 * cells;
  const float* bi = fm + 4 * cells;
  const float* bd = fm + 5 * cells;
  float* probs = out + pr.out_off;
  const size_t last = static_cast<size_t>(n1 - 2) * n2 + (n2 - 2);
  float global_sum = fm[last];
  global_sum = lse(global_sum, fi[last], tab);
  global_sum = lse(global_sum, fd[last], tab);
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t x = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; x < cells; x += stride) {
    const uint32_t i = static_cast<uint32_t>(x / n2), j = static_cast<uint32_t>(x - static_cast<size_t>(i) * n2);
    float p = 0.f;
    if (i >= 1 && i + 1 < n1 && j >= 1 && j + 1 < n2) {
      const size_t nx = static_cast<size_t>(i + 1) * n2 + (j + 1);
      const bool ends = (i + 1 == n1 - 1) && (j + 1 == n2 - 1);
      float sum = kNegInf;
      sum = lse(sum, (ends ? 0.f : sc.match2match_score) + bm[nx], tab);
      sum = lse(sum, sc.match2insert_score + bi[nx], tab);
      sum = lse(sum, sc.match2insert_score + bd[nx], tab);
      p = expf_ref(fm[x] + sum - global_sum);
    }
    probs[x] = p;
  }
}

// one wave per listed cell: closing-pair block of diagonal d (inside) / 2-loop half of the pair
// probability of diagonal d (outside), rnamc_latency.h
template <bool CONTRA, bool OUTSIDE>
__global__ void __launch_bounds__(64) k_pair_lat(DeviceBatch b, uint32_t d, uint32_t nseq) {
  const uint32_t bx = blockIdx.x / nseq;
  const uint32_t which = blockIdx.x - bx * nseq;
  const Seq q = load_seq(b, which);
  const uint32_t n = q.n;
  if (d >= n) return;
  const uint32_t cnt = q.ccnt[d];
  if (bx >= cnt) return;
  const uint32_t i = uni(static_cast<uint32_t>(q.cidx[tri_off(n, d) + bx]));
  const Piece8 P8 = load_piece8();
  if (OUTSIDE) {
    outside_head_lat<CONTRA>(b, q, d, i, P8);
  } else {
    inside_pair_lat<CONTRA, PAIR_FULL>(b, q, d, i, P8);
  }
}

// final map (src/mccaskill_algo.rs:608 / 721) + log partition function.  A pair is
// in the reference's SparseProbMat iff it got a probability, i.e. iff it is in
// sums_close and its span was visited by the outside sweep (602-604 / 715-717).
__global__ void k_finalize(DeviceBatch b, uint32_t dmin_out) {
  const SeqDesc sd = b.seqs[blockIdx.y];
  float* out = b.out + sd.out_off;
  const float* base = b.workspace + sd.ws_off;
  const float* lp = base + static_cast<size_t>(M_P) * sd.tri_pad;
  const float* qb = base + static_cast<size_t>(M_QB) * sd.tri_pad;
  const size_t olen = static_cast<size_t>(sd.n) * (sd.n + 1u) / 2u;
  const size_t first = tri_off(sd.n, min(dmin_out, sd.n));  // cells of shorter spans
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t x = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; x < olen;
       x += stride) {
    const bool present = x >= first && qb[x] > kNegInf;
    out[x] = present ? expf_ref(lp[x]) : -1.0f;
  }
  if (b.log_partition && blockIdx.x == 0 && threadIdx.x == 0) {
    const float* z = base + static_cast<size_t>(M_Z) * sd.tri_pad;
    b.log_partition[sd.batch_idx] = z[tri_off(sd.n, sd.n - 1)];
  }
}

}  // namespace

// ----------------------------------------------------------------------------
// launch wrappers (host)

void launch_init(const DeviceBatch& b, uint32_t nseq, uint32_t max_n, hipStream_t st) {
  const uint64_t elems = static_cast<uint64_t>(max_n) * (max_n + 1) / 2 * M_COUNT;
  uint32_t gx = static_cast<uint32_t>(std::min<uint64_t>((elems + 1023) / 1024, 512));
  if (gx == 0) gx = 1;
  hipLaunchKernelGGL(k_init, dim3(gx, nseq, 1), dim3(256), 0, st, b);
  hipLaunchKernelGGL(k_compact, dim3(max_n, nseq, 1), dim3(64), 0, st, b);
}

// sums of diagonal d (if do_sums) and closing-pair block of diagonal d+1 (if do_pair)
void launch_inside(const DeviceBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                   uint32_t block, bool do_sums, bool do_pair, hipStream_t st) {
  const uint32_t cells_s = (do_sums && d < max_n) ? max_n - d : 0;
  const uint32_t cells_p = (do_pair && d + 1 < max_n) ? max_n - d - 1 : 0;
  // launches that cannot give every SIMD a wave are latency-bound: spread the three
  // chains of a cell over three lanes (21 cells per wave)
  const bool split = do_sums && inside_is_split(d, max_n, nseq);
  const uint32_t cells_per_block = split ? (block / 64) * kSplitCells : block;
  const uint32_t bs = (cells_s + cells_per_block - 1) / cells_per_block;
  const uint32_t bp = (cells_p + block - 1) / block;
  if (bs + bp == 0 || nseq == 0) return;
  const dim3 g((bs + bp) * nseq, 1, 1);
  const int ds = do_sums ? 1 : 0, dp = do_pair ? 1 : 0;
  if (contra) {
    if (split) {
      hipLaunchKernelGGL((k_inside<true, true>), g, dim3(block), 0, st, b, d, bs, nseq, ds, dp);
    } else {
      hipLaunchKernelGGL((k_inside<true, false>), g, dim3(block), 0, st, b, d, bs, nseq, ds, dp);
    }
  } else {
    if (split) {
      hipLaunchKernelGGL((k_inside<false, true>), g, dim3(block), 0, st, b, d, bs, nseq, ds, dp);
    } else {
      hipLaunchKernelGGL((k_inside<false, false>), g, dim3(block), 0, st, b, d, bs, nseq, ds, dp);
    }
  }
}

// folds of diagonals d and d+1 (d >= 2; both closing-pair blocks done) and, if do_head, the
// early part of the closing-pair blocks of diagonals d+2 and d+3
void launch_inside2(const DeviceBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                    uint32_t block, bool do_sums, bool do_head, hipStream_t st) {
  if (d >= max_n || nseq == 0) return;
  const uint32_t bs = do_sums ? (max_n - d + block - 1) / block : 0;
  const uint32_t cells_h = (do_head && d + 2 < max_n) ? max_n - d - 2 : 0;
  const uint32_t bh = std::max(1u, (cells_h + block - 1) / block);
  const uint32_t nh = cells_h ? 2u : 0u;
  if (bs + nh == 0) return;
  const dim3 g((bs + nh * bh) * nseq, 1, 1);
  const int ds = do_sums ? 1 : 0, dh = do_head ? 1 : 0;
  if (contra) {
    if (nh) hipLaunchKernelGGL((k_inside2<true, false, true>), g, dim3(block), 0, st, b, d, bs, bh, nseq, ds, dh);
    else hipLaunchKernelGGL((k_inside2<true, false, false>), g, dim3(block), 0, st, b, d, bs, bh, nseq, ds, dh);
  } else {
    if (nh) hipLaunchKernelGGL((k_inside2<false, false, true>), g, dim3(block), 0, st, b, d, bs, bh, nseq, ds, dh);
    else hipLaunchKernelGGL((k_inside2<false, false, false>), g, dim3(block), 0, st, b, d, bs, bh, nseq, ds, dh);
  }
}

// CONTRAfold: the rightmost-pair sums of diagonals d and d+1 (precede launch_inside2)
void launch_inside_zr2(const DeviceBatch& b, uint32_t d, uint32_t max_n, uint32_t nseq,
                       uint32_t block, hipStream_t st) {
  if (d >= max_n || nseq == 0) return;
  const uint32_t bs = (max_n - d + block - 1) / block;
  hipLaunchKernelGGL((k_inside2<true, true, false>), dim3(bs * nseq, 1, 1), dim3(block), 0, st, b, d,
                     bs, 1u, nseq, 1, 0);
}

// multibranch term of the parked closing-pair blocks of diagonals d0 .. d0+nd-1
void launch_pair_tail(const DeviceBatch& b, bool contra, uint32_t d0, uint32_t nd, uint32_t max_n,
                      uint32_t nseq, uint32_t block, hipStream_t st) {
  if (d0 >= max_n || nseq == 0 || nd == 0) return;
  const uint32_t bd = (max_n - d0 + block - 1) / block;
  const dim3 g(nd * bd * nseq, 1, 1);
  if (contra) {
    hipLaunchKernelGGL(k_pair_tail<true>, g, dim3(block), 0, st, b, d0, bd, nseq);
  } else {
    hipLaunchKernelGGL(k_pair_tail<false>, g, dim3(block), 0, st, b, d0, bd, nseq);
  }
}

bool inside_is_split(uint32_t d, uint32_t max_n, uint32_t nseq) {
  const uint32_t cells_s = d < max_n ? max_n - d : 0;
  return static_cast<uint64_t>(cells_s) * nseq < 64ull * 1024ull;
}

// launch of diagonal d: probs_multibranch (do_mb) and multibranch half of the pair
// probabilities (do_tail) of diagonal d, 2-loop half (do_head) of diagonal d-1
void launch_outside(const DeviceBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                    uint32_t block, bool do_mb, bool do_tail, bool do_head, int roles,
                    hipStream_t st) {
  if (nseq == 0) return;
  const uint32_t nb = d < max_n ? (max_n - d + block - 1) / block : 0;
  const uint32_t nh = (do_head && d >= 1 && d - 1 < max_n) ? (max_n - d + 1 + block - 1) / block : 0;
  if (2 * nb + nh == 0) return;
  const dim3 g((2 * nb + nh) * nseq, 1, 1);
  const int a0 = do_mb ? 1 : 0, a1 = do_tail ? 1 : 0, a2 = do_head ? 1 : 0;
#define RNAMC_LAUNCH_OUT(C, R) \
  hipLaunchKernelGGL((k_outside<C, R>), g, dim3(block), 0, st, b, d, nb, nh, nseq, a0, a1, a2)
  if (contra) {
    if (roles == 5) RNAMC_LAUNCH_OUT(true, 5);
    else if (roles == 4) RNAMC_LAUNCH_OUT(true, 4);
    else if (roles == 1) RNAMC_LAUNCH_OUT(true, 1);
    else if (roles == 2) RNAMC_LAUNCH_OUT(true, 2);
    else RNAMC_LAUNCH_OUT(true, 7);
  } else {
    if (roles == 5) RNAMC_LAUNCH_OUT(false, 5);
    else if (roles == 4) RNAMC_LAUNCH_OUT(false, 4);
    else if (roles == 1) RNAMC_LAUNCH_OUT(false, 1);
    else if (roles == 2) RNAMC_LAUNCH_OUT(false, 2);
    else RNAMC_LAUNCH_OUT(false, 7);
  }
#undef RNAMC_LAUNCH_OUT
}

// latency forms (rnamc_latency.h): the fold chains of diagonal d (form 1: one wave per chain,
// 2: eight chains per wave, 0: none) and the combine that completes sums_1ormore_basepairs of
// diagonal d-1 (do_combine)
void launch_inside_lat(const DeviceBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                       int form, bool do_combine, uint32_t pair_d, hipStream_t st) {
  if (nseq == 0) return;
  const uint32_t cells = ((form & 3) && d < max_n) ? max_n - d : 0;
  const uint32_t per = 8u;
  const uint32_t waves = ((form & 4) ? 5u : 3u) * ((cells + per - 1u) / per);
  const uint32_t cb = (do_combine && d >= 1 && d - 1 < max_n) ? (max_n - d + 1 + 63) / 64 : 0;
  const uint32_t pc = (pair_d != 0 && pair_d < max_n) ? max_n - pair_d : 0;  // pair_d 0: none
  if (waves + cb + pc == 0) return;
  const dim3 g((pc + waves + cb) * nseq, 1, 1);
  const int a1 = do_combine ? 1 : 0;
  if (contra) {
    hipLaunchKernelGGL(k_inside_lat<true>, g, dim3(64), 0, st, b, d, cells, nseq, cells ? form : 0, a1,
                       pair_d, pc);
  } else {
    hipLaunchKernelGGL(k_inside_lat<false>, g, dim3(64), 0, st, b, d, cells, nseq, cells ? form : 0, a1,
                       pair_d, pc);
  }
}

// probs_multibranch{,2} and the multibranch half of the pair probabilities of diagonal d, one
// wave per cell / listed cell
// (head: with the 2-loop half of the pair probabilities of diagonal d - 1 in the same launch)
void launch_outside_lat(const DeviceBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                        bool do_mb, bool do_tail, bool head, hipStream_t st) {
  if (nseq == 0) return;
  const uint32_t cells = d < max_n ? max_n - d : 0;
  const uint32_t head_d = d >= 1 ? d - 1 : 0;
  const uint32_t hc = (head && d >= 1 && head_d < max_n) ? max_n - head_d : 0;
  if (cells + hc == 0) return;
  const dim3 g((2 * cells + hc) * nseq, 1, 1);
  const int a0 = do_mb ? 1 : 0, a1 = do_tail ? 1 : 0;
  if (contra) {
    hipLaunchKernelGGL(k_outside_lat<true>, g, dim3(64), 0, st, b, d, cells, nseq, a0, a1, head_d, hc);
  } else {
    hipLaunchKernelGGL(k_outside_lat<false>, g, dim3(64), 0, st, b, d, cells, nseq, a0, a1, head_d, hc);
  }
}

void launch_durbin(const DurbinPair* d_pairs, uint32_t n_pairs, uint32_t max_cells,
                   const uint8_t* d_bases, float* ws, float* d_out, const rnamc_align_scores& sc,
                   hipStream_t st) {
  if (n_pairs == 0) return;
  hipLaunchKernelGGL(k_durbin_sums, dim3(2 * n_pairs), dim3(kDurbinThreads), 0, st, d_pairs, d_bases,
                     ws, sc);
  const uint32_t gx = std::max<uint32_t>(1u, std::min<uint32_t>((max_cells + 255u) / 256u, 256u));
  hipLaunchKernelGGL(k_durbin_probs, dim3(gx, n_pairs), dim3(256), 0, st, d_pairs, ws, d_out, sc);
}

// latency form of the 2-loop blocks: one wave per listed cell of diagonal d
void launch_pair_lat(const DeviceBatch& b, bool contra, bool outside, uint32_t d, uint32_t max_n,
                     uint32_t nseq, hipStream_t st) {
  if (d >= max_n || nseq == 0) return;
  const dim3 g((max_n - d) * nseq, 1, 1);
  if (contra) {
    if (outside) hipLaunchKernelGGL((k_pair_lat<true, true>), g, dim3(64), 0, st, b, d, nseq);
    else hipLaunchKernelGGL((k_pair_lat<true, false>), g, dim3(64), 0, st, b, d, nseq);
  } else {
    if (outside) hipLaunchKernelGGL((k_pair_lat<false, true>), g, dim3(64), 0, st, b, d, nseq);
    else hipLaunchKernelGGL((k_pair_lat<false, false>), g, dim3(64), 0, st, b, d, nseq);
  }
}

void launch_finalize(const DeviceBatch& b, uint32_t nseq, uint32_t max_n, uint32_t dmin_out,
                     hipStream_t st) {
  const uint64_t elems = static_cast<uint64_t>(max_n) * (max_n + 1) / 2;
  uint32_t gx = static_cast<uint32_t>(std::min<uint64_t>((elems + 1023) / 1024, 256));
  if (gx == 0) gx = 1;
  hipLaunchKernelGGL(k_finalize, dim3(gx, nseq, 1), dim3(256), 0, st, b, dmin_out);
}

}  // namespace rnamc

#ifdef RNAMC_COUNT_FAR
// counting build only: copy (and optionally clear) the tallies of FarCount
extern "C" int rnamc_debug_far_counters(unsigned long long* out32, int reset) {
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(rnamc::g_far_counters), sizeof(unsigned long long) * 32) != hipSuccess)
    return 7;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(rnamc::g_far_counters), z, sizeof(z)) != hipSuccess) return 7;
  }
  return 0;
}
#endif
