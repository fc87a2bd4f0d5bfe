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

#include <algorithm>
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
// Wave-uniform loads through the scalar unit (s_load): the operand sits at an address every lane
// of the wave shares, and it was written by an EARLIER launch (the scalar cache is invalidated
// at every kernel start like the vector L1).  A wave's uniform operands then cost the vector
// memory pipeline nothing, which matters: a launch runs thousands of waves that each need ~40.
__device__ __forceinline__ float sload(const float* p) {
  return *reinterpret_cast<const __attribute__((address_space(4))) float*>(reinterpret_cast<uintptr_t>(p));
}
__device__ __forceinline__ uint32_t sload(const uint32_t* p) {
  return *reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(reinterpret_cast<uintptr_t>(p));
}
__device__ __forceinline__ float2 sload2(const float2* p) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f v = *reinterpret_cast<const __attribute__((address_space(4))) v2f*>(reinterpret_cast<uintptr_t>(p));
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ float4 sload4(const float4* p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = *reinterpret_cast<const __attribute__((address_space(4))) v4f*>(reinterpret_cast<uintptr_t>(p));
  return make_float4(v.x, v.y, v.z, v.w);
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
__device__ __forceinline__ bool cell_reduce(Acc (&a)[NA], float (*lds)[NA][2], uint32_t livemask = (1u << NA) - 1u) {
  // (`livemask`, wave-uniform: bit x clear = accumulator x is empty in every lane of the group)
#pragma unroll
  for (int x = 0; x < NA; x++)
    if ((livemask >> x) & 1u) a[x] = wave_reduce(a[x]);
  if (TPC == 64) return true;
  constexpr int W = TPC / 64;  // waves per cell group (a 256-thread block holds 256 / TPC groups)
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t gw = wave % static_cast<uint32_t>(W), first = wave - gw;
  if (lane == 0u) {
#pragma unroll
    for (int x = 0; x < NA; x++)
      if ((livemask >> x) & 1u) {
        lds[wave][x][0] = a[x].m;
        lds[wave][x][1] = a[x].s;
      }
  }
  __syncthreads();
  if (gw != 0u) return false;
#pragma unroll
  for (int x = 0; x < NA; x++)
    if ((livemask >> x) & 1u) {
      Acc t = lane < static_cast<uint32_t>(W) ? Acc{lds[first + lane][x][0], lds[first + lane][x][1]} : acc_empty();
      a[x] = wave_reduce(t);
    }
  return true;
}

// (+)_k (A[k] + B[k]) over k in [0, len): both operands contiguous in k, the cell's TPC
// lanes take consecutive k (256-B wave accesses).  The sweep's launches are bound by how many
// bytes their waves keep in flight (a wave that waits for one load per step streams at a few
// per cent of the HBM rate): every lane issues the loads of FOUR steps of every stream before
// it folds the first.
template <int TPC>
__device__ __forceinline__ void acc_product(Acc& a, const float* __restrict__ A,
                                            const float* __restrict__ B, uint32_t len, uint32_t t) {
  for (uint32_t k = t; k < len; k += 8u * TPC) {
    // (all loads of the chunk first, unconditional — index clamped to the stream's first element,
    // validity applied to the value — and a scheduling barrier before the arithmetic: left alone the
    // compiler issued two loads, waited, summed, two loads, waited ...: eight round trips per chunk)
    float x[8], va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t ku = k + static_cast<uint32_t>(u) * TPC;
      const uint32_t kc = ku < len ? ku : 0u;
      va[u] = A[kc];
      vb[u] = B[kc];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; u++) x[u] = (k + static_cast<uint32_t>(u) * TPC < len) ? va[u] + vb[u] : kNegInf;
    acc_add4(a, x[0], x[1], x[2], x[3]);
    acc_add4(a, x[4], x[5], x[6], x[7]);
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
  float2* mid;   // banded mid-field ring: [product][d % ring][i] = {max, sum} (see k_tree_mid)
  float2* far;   // far parts of the next launch's 2-loop blocks: [d & 3][i] = {max, sum} (see Ahead)
  uint32_t vec;  // cells per ring row
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
  q.mid = reinterpret_cast<float2*>(b.workspace + sd.mid_off);
  q.vec = (sd.n + 64u + 63u) & ~63u;
  q.far = q.mid + static_cast<size_t>(3u) * b.ring * q.vec;
  return q;
}

__device__ __forceinline__ uint32_t tri_off(uint32_t n, uint32_t d) {
  return d * n - (d * (d - 1u)) / 2u;
}

// The sweep's launches take what their FIRST loads are addressed with — the lone sequence's matrix
// base, matrix size, n and row stride, the diagonal, the band threshold — as leading SCALAR kernel
// arguments: with -mllvm -amdgpu-kernarg-preload-count the command processor places those in SGPRs
// at dispatch, so a wave's ~45 operand loads need not wait for its own read of the kernel
// argument segment (a dependent scalar round trip in front of every launch of the chain).
__device__ __forceinline__ TSeq load_tseq_hot(const TreeBatch& b, uint32_t which, float* hbase, uint64_t hmsz,
                                              uint32_t hn, uint32_t hld, uint32_t use_one) {
  if (!use_one) return load_tseq(b, which);
  const TreeSeq& sd = b.one;
  TSeq q;
  q.s = b.bases + sd.seq_off;
  q.n = hn;
  q.ld = hld;
#pragma unroll
  for (int x = 0; x < T_COUNT; x++) q.m[x] = hbase + static_cast<size_t>(x) * hmsz;
  q.zp = hbase + static_cast<size_t>(T_COUNT) * hmsz;
  q.zs = q.zp + (hn + 64u);
  q.pk = reinterpret_cast<const uint32_t*>(b.workspace + sd.pk_off);
  q.out = b.out + sd.out_off;
  q.batch_idx = sd.batch_idx;
  q.mid = reinterpret_cast<float2*>(b.workspace + sd.mid_off);
  q.vec = (hn + 64u + 63u) & ~63u;
  q.far = q.mid + static_cast<size_t>(3u) * b.ring * q.vec;
  return q;
}

// 32 consecutive bases p0 .. p0+31 in one 64-bit value (window position q at bits 2q, 2q+1);
// p0 >= -32 (the packed copy carries 32 zero bases in front and >= 64 behind)
__device__ __forceinline__ uint64_t load_win64(const uint32_t* __restrict__ pk, int p0) {
  const uint32_t bit = 2u * static_cast<uint32_t>(p0 + 32);
  const uint32_t w = bit >> 5, sh = bit & 31u;
  const uint32_t w0 = sload(pk + w), w1 = sload(pk + w + 1), w2 = sload(pk + w + 2);  // (p0 is uniform)
  const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh);
  const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sh);
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
__device__ __forceinline__ int wb(uint64_t w, uint32_t q) { return static_cast<int>((w >> (2u * q)) & 3u); }
// the same window at a per-lane position (vector loads)
__device__ __forceinline__ uint64_t load_win64v(const uint32_t* __restrict__ pk, int p0) {
  const uint32_t bit = 2u * static_cast<uint32_t>(p0 + 32);
  const uint32_t w = bit >> 5, sh = bit & 31u;
  const uint32_t w0 = pk[w], w1 = pk[w + 1], w2 = pk[w + 2];
  const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh);
  const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sh);
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
// A 2-loop (i,j) around (k,l) = (i+1+a, j-1-b) reads a pair a + b + 2 diagonals below its own.  With
// two diagonals per launch the pairs of the last three diagonals are the only ones a launch's
// blocks cannot have seen one launch earlier: the slots with a + b <= 1, i.e. the first kNear
// explicit small loops of both models ((0,0) (0,1) (1,0)).  Everything else — the <= 493 FAR
// slots, all of a block's gathers — is summed by extra workgroups of the PREVIOUS launch
// (blockIdx.x >= main_blocks; they depend on nothing that launch computes, so they fill the
// issue slots its chain of round trips leaves idle, and each cell's far part is formed once
// instead of once per group that needs it) and handed over as a {max, sum} pair.  The outside
// sweep mirrors it: (k,l) = (i-1-a, j+1+b) lies a + b + 2 diagonals above.
constexpr uint32_t kNear = 3u;
#ifndef RNAMC_FAR_NG
#define RNAMC_FAR_NG 8
#endif
constexpr int kFarNG = RNAMC_FAR_NG;  // gathers per lane in flight in the far parts (8 slots per lane)
struct Ahead {
  uint32_t flags;        // bit 0: this launch's blocks take their far part from q.far; bit 1: one ahead
                         // wave per CELL; bit 2: workgroup ids dealt by xcd_chunk
  uint32_t main_blocks;  // workgroups of the launch proper; the rest are the ahead role
  uint32_t nd0, nd_count;  // diagonals of the next launch (nd_count = 0: no ahead role)
};

// Workgroups go to the 8 XCDs round-robin by linear id (x fastest), so with 4 rows to a workgroup
// neighbouring rows of a sequence — which read the same lines: the 31 x 31 window of a closing
// pair's 2-loops, the product columns — meet in no L2.  Flag bit 2: both roles' x ranges are
// padded to multiples of 64 (so that x mod 8 IS the XCD whatever the sequence) and x is dealt so
// that 8 workgroups in a row, 32 rows, land on one XCD, the XCDs rotated by sequence (short
// sequences fill only the first runs: without the rotation those would always be XCD 0's).
__device__ __forceinline__ uint32_t xcd_chunk(uint32_t x, uint32_t y) {
  const uint32_t c = x & 7u, k = x >> 3;
  return ((k & ~7u) + ((c + y) & 7u)) * 8u + (k & 7u);
}
__device__ __forceinline__ uint32_t role_block(const Ahead& ah) {
  uint32_t bx = blockIdx.x;
  if (ah.flags & 4u)
    bx = bx < ah.main_blocks ? xcd_chunk(bx, blockIdx.y) : ah.main_blocks + xcd_chunk(bx - ah.main_blocks, blockIdx.y);
  return bx;
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
    // (the statics CS4 / IN4 / ACCS are read only where MBC says "may pair": any finite filler)
    const size_t total = static_cast<size_t>(T_COUNT) * sd.msz;
    if (b.lane) {
      // (lane-per-cell sweeps: k_tree_static writes MBC — the "may pair" flag — for every cell and the
      // other statics where it is finite, their only readers; the lists and the generic 2-loop sums are written whole: 22 of the 38 slots need no filler)
      for (int mi = 0; mi < T_COUNT; mi++) {
        if (mi == T_HP || mi == T_MBC || (mi >= T_ACCS && mi < T_QB_D) || mi >= T_LIST) continue;  // (.. T_LIST, T_GEN_D's two slots)
        float* p = base + static_cast<size_t>(mi) * sd.msz;
        for (size_t x = t0; x < sd.msz; x += stride) p[x] = kNegInf;
      }
    } else
    for (size_t x = t0; x < total; x += stride) {
      const size_t mi = x / sd.msz;
      base[x] = (mi >= T_ACCS && mi < T_QB_D) ? 0.f : kNegInf;
    }
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
    // the slots the outside sweep reuses (W, R, Pm2, SP, and PX4 in place of QbX4)
    // (lane-per-cell sweeps: R takes the diagonal-major Zr_mb's place too)
    const int mats[10] = {T_ZRE, T_ZRM, T_QM, T_U, T_X4, T_X4 + 1, T_X4 + 2, T_X4 + 3, T_ZRM_D, T_W_D};
    for (int y = 0; y < (b.lane ? 10 : 8); y++) {
      float* p = base + static_cast<size_t>(mats[y]) * sd.msz;
      for (size_t x = t0; x < sd.msz; x += stride) p[x] = kNegInf;
    }
  }
}

// ----------------------------------------------------------------------------
// Per-cell statics (functions of the sequence alone), written once per sequence by
// k_tree_static so that the sweep's launches find them with ONE load each: a launch's duration
// is the length of its waves' chain of dependent memory round trips, and scores looked up
// through base codes (bases -> table) would add two levels to it.
//   HP, MBC, ACCS : hairpin, multibranch-close and accessible score of the pair (i,j); MBC is
//                   -inf where (i,j) may not pair (the `act` flag of the sweep)
//   CS4(i,j)      : the pair as the CLOSING pair of a generic 2-loop, by class: Turner
//                   pen | mismatch_c[si][sj][s(i+1)][s(j-1)] + pen; CONTRAfold helix_close +
//                   terminal_mismatch
//   IN4(i,j)      : the pair as the ENCLOSED pair: mismatch_c[sj][si][s(j+1)][s(i-1)] + pen
//                   (CONTRAfold: + base-pair score)
// A generic 2-loop (i,j) around (k,l) then scores  len[slot] + CS4(i,j)[c] + IN4(k,l)[c]  with
// c = class of the slot; the sweep stores  QbX4(k,l) = sums_close(k,l) + IN4(k,l)  (inside) and
// PX4(k,l) = (log bpp - sums_close)(k,l) + CS4(k,l)  (outside) as float4 per cell, so a probe is
// one 16-byte gather.  The few small loops with explicit tables (Turner: stack, 0x1, 1x1, 1x2,
// 2x1, 2x2; CONTRAfold: stack, 0x1, 1x1) depend on both pairs at once: the flat scorers of
// rnamc_scoring.h score them in one extra pass of the cell's first lanes.
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
  static __device__ __forceinline__ bool is(uint32_t a, uint32_t bb) {
    return (a + bb <= 1u) || (a >= 1u && a <= 2u && bb >= 1u && bb <= 2u);
  }
};
template <>
struct Special<true> {
  static constexpr uint32_t N = 4;
  static __device__ __forceinline__ void slot(uint32_t t, uint32_t& a, uint32_t& bb) {
    a = t >> 1;  // (0,0) (0,1) (1,0) (1,1)
    bb = t & 1u;
  }
  static __device__ __forceinline__ bool is(uint32_t a, uint32_t bb) { return a <= 1u && bb <= 1u; }
};

// class of a generic slot: 0 bulge, 1 1 x many, 2 2 x 3, 3 other interior
__device__ __forceinline__ uint32_t slot_class(uint32_t a, uint32_t bb) {
  if ((a == 0u) != (bb == 0u)) return 0u;
  if (a == 1u || bb == 1u) return 1u;
  if ((a == 2u && bb == 3u) || (a == 3u && bb == 2u)) return 2u;
  return 3u;
}
__device__ __forceinline__ float pick(const float4& v, uint32_t c) {
  return c == 0u ? v.x : (c == 1u ? v.y : (c == 2u ? v.z : v.w));
}

template <bool CONTRA>
__device__ __forceinline__ float4 fixed_side(const TreeBatch& b, int p0, int p1, int q0, int q1, float extra) {
  if (CONTRA) {
    const rnamc_fold_score_sets& f = b.params->contra;
    const float x = (f.helix_close_scores[p0][p1] + f.terminal_mismatch_scores[p0][p1][q0][q1]) + extra;
    return make_float4(x, x, x, x);
  }
  const rnamc_turner_scores& tt = b.params->turner;
  const float pen = augu(p0, p1) ? tt.helix_augu_end_penalty : 0.f;
  return make_float4(pen, tt.terminal_mismatch_scores_1xmany[p0][p1][q0][q1] + pen,
                     tt.terminal_mismatch_scores_2x3[p0][p1][q0][q1] + pen,
                     tt.terminal_mismatch_scores_interior[p0][p1][q0][q1] + pen);
}

template <bool CONTRA>
__global__ void __launch_bounds__(256) k_tree_static(TreeBatch b) {
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  const uint8_t* __restrict__ s = q.s;
  const auto model = TModel<CONTRA>::make(b);
  // row-major statics: x = i n + j; diagonal-major (lane-per-cell sweeps): x = d ld + i, so that a wave's
  // stores are consecutive, and MBC is written for EVERY cell (k_tree_init leaves the statics alone there)
  const uint64_t cells = static_cast<uint64_t>(n) * (b.lane ? ld : n);
  for (uint64_t x = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; x < cells;
       x += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    uint32_t i, j;
    if (b.lane) {
      const uint32_t x32 = static_cast<uint32_t>(x);  // (n ld < 2^32 for every admitted n)
      const uint32_t dd = x32 / ld;
      i = x32 - dd * ld;
      j = i + dd;
      if (j >= n) continue;
      if (dd == 0u) {
        q.m[T_MBC][x32] = kNegInf;
        continue;
      }
    } else {
      i = static_cast<uint32_t>(x / n);
      j = static_cast<uint32_t>(x % n);
      if (j <= i) continue;
    }
    const uint32_t d = j - i;
    const int si = s[i], sj = s[j];
    const bool act = canonical(si, sj) &&
                     ((b.allows_short_hairpins && CONTRA) || d + 1 >= RNAMC_MIN_SPAN_HAIRPIN_CLOSE);
    const size_t o = b.lane ? static_cast<size_t>(d) * ld + i : static_cast<size_t>(i) * ld + j;
    if (!act) {  // (row-major: the slots were filled with -inf / 0 by k_tree_init)
      if (b.lane) q.m[T_MBC][o] = kNegInf;
      continue;
    }
    q.m[T_HP][o] = (!CONTRA || d - 1 <= RNAMC_MAX_LOOP_LEN) ? model.hairpin(s, n, i, j) : kNegInf;
    q.m[T_MBC][o] = model.mbclose(s, n, i, j);
    q.m[T_ACCS][o] = model.accessible(s, n, i, j);
    reinterpret_cast<float4*>(q.m[T_CS4])[o] = fixed_side<CONTRA>(b, si, sj, s[i + 1], s[j - 1], 0.f);
    float4 in4 = make_float4(0.f, 0.f, 0.f, 0.f);  // (a pair at either end encloses nothing)
    if (i >= 1 && j + 1 < n)
      in4 = fixed_side<CONTRA>(b, sj, si, s[j + 1], s[i - 1],
                               CONTRA ? b.params->contra.basepair_scores[si][sj] : 0.f);
    reinterpret_cast<float4*>(q.m[T_IN4])[o] = in4;
    // the three nearest explicit 2-loops this pair closes (slots (0,0) (0,1) (1,0): kNear), so
    // that the sweep's launches find their scores with the cell's other operands
    // (all of them where the lane-per-cell sweeps run: no scorer and no base windows in those)
    float nr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint32_t nsl = b.lane ? Special<CONTRA>::N : kNear;
    for (uint32_t t = 0; t < nsl; t++) {
      uint32_t a, bb;
      Special<CONTRA>::slot(t, a, bb);
      if (a + bb + 3u <= d)
        nr[t] = TModel<CONTRA>::twoloop(b, a, bb, si, sj, s[i + 1], s[i + 2], s[j - 1], s[j - 2], s[i + 1 + a],
                                        s[j - 1 - bb], s[j - bb], s[i + a]);
    }
    reinterpret_cast<float4*>(q.m[T_NEAR4])[o] = make_float4(nr[0], nr[1], nr[2], nr[3]);
    if (b.lane && Special<CONTRA>::N > 4) reinterpret_cast<float4*>(q.m[T_NEAR8])[o] = make_float4(nr[4], nr[5], nr[6], 0.f);
  }
}

// closing-pair block of the pair (i,j) (src/mccaskill_algo.rs:297-343 / 400-467): hairpin and
// multibranch term (held by the caller, uniform) and the <= 496 enclosed pairs
// (k,l) = (i+1+a, j-1-b); wi = bases i .. i+31, wj = bases j-31 .. j
template <bool CONTRA, int TPC>
__device__ __forceinline__ void pair_block(const TreeBatch& b, const TSeq& q, Acc& acc, uint32_t i,
                                           uint32_t j, uint32_t t, float hp, float mbt,
                                           const float4& cs4, uint64_t wi, uint64_t wj) {
  const uint32_t d = j - i, ld = q.ld;
  acc_add(acc, t == 0u ? hp : (t == 1u ? mbt : kNegInf));
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 1) return;
#endif
  if (d < 3u) return;
  const float* __restrict__ qx = q.m[T_X4];  // four planes, one per class
  const size_t msz = static_cast<size_t>(q.m[1] - q.m[0]);
  const float* __restrict__ tlen = &b.tabs->len[CONTRA ? 1 : 0][0];
  constexpr int NP = (512 + TPC - 1) / TPC;
  float g[NP];
  float ln[NP];
  uint32_t cls[NP];
  // every gather of the block first (one round trip), then the sums
#pragma unroll
  for (int u = 0; u < NP; u++) {
    const uint32_t p = t + static_cast<uint32_t>(u) * TPC;
    uint32_t a = 0, bb = 0;
    const bool ok = p < 512u && probe_slot(p, a, bb) && !Special<CONTRA>::is(a, bb) && a + bb + 3u <= d;
    cls[u] = slot_class(a, bb);
    ln[u] = ok ? tlen[p] : kNegInf;
    g[u] = ok ? qx[cls[u] * msz + static_cast<size_t>(i + 1u + a) * ld + (j - 1u - bb)] : 0.f;
  }
  float sx = kNegInf;  // the lane's small explicit loop, if it has one
  if (t < Special<CONTRA>::N) {
    uint32_t a, bb;
    Special<CONTRA>::slot(t, a, bb);
    if (a + bb + 3u <= d) {
      // (k,l) at window positions 1+a / 30-bb, their outer neighbours at a / 31-bb
      const uint32_t k = i + 1u + a, l = j - 1u - bb;
      const float x = q.m[T_QB][static_cast<size_t>(k) * ld + l];
      const float sc = TModel<CONTRA>::twoloop(b, a, bb, wb(wi, 0), wb(wj, 31), wb(wi, 1), wb(wi, 2),
                                               wb(wj, 30), wb(wj, 29), wb(wi, 1u + a), wb(wj, 30u - bb),
                                               wb(wj, 31u - bb), wb(wi, a));
      sx = x + sc;  // (absent pair: x = -inf)
    }
  }
  // lane-local two-pass sum (one exp2 per term), merged into the accumulator once
  float xv[NP], mx = sx;
#pragma unroll
  for (int u = 0; u < NP; u++) {
    xv[u] = (g[u] + ln[u]) + pick(cs4, cls[u]);
    mx = vmaxf(mx, xv[u]);
  }
  mx = vmaxf(mx, kEmpty);
  float sm = ex2((sx - mx) * kL2E);
#pragma unroll
  for (int u = 0; u < NP; u++) sm += ex2((xv[u] - mx) * kL2E);
  acc_merge(acc, Acc{mx, sm});
}

// The FAR part of the same block alone (ahead role, one wave per cell): no hairpin / multibranch
// term, explicit small loops from slot kNear on.  Off the launch's critical path, so the gathers
// go four per lane at a time (two round trips, half the registers: the kernel stays resident
// eight waves per SIMD deep).
template <bool CONTRA>
__device__ __forceinline__ void pair_far(const TreeBatch& b, const TSeq& q, Acc& acc, uint32_t i, uint32_t j,
                                         uint32_t t, const float4& cs4, uint64_t wi, uint64_t wj) {
  const uint32_t d = j - i, ld = q.ld;
  if (d < 3u) return;
  const float* __restrict__ qx = q.m[T_X4];  // four planes, one per class
  const size_t msz = static_cast<size_t>(q.m[1] - q.m[0]);
  const float* __restrict__ tlen = &b.tabs->len[CONTRA ? 1 : 0][0];
  float sx = kNegInf;
  if (t >= kNear && t < Special<CONTRA>::N) {
    uint32_t a, bb;
    Special<CONTRA>::slot(t, a, bb);
    if (a + bb + 3u <= d) {
      const uint32_t k = i + 1u + a, l = j - 1u - bb;
      const float x = q.m[T_QB][static_cast<size_t>(k) * ld + l];
      const float sc = TModel<CONTRA>::twoloop(b, a, bb, wb(wi, 0), wb(wj, 31), wb(wi, 1), wb(wi, 2),
                                               wb(wj, 30), wb(wj, 29), wb(wi, 1u + a), wb(wj, 30u - bb),
                                               wb(wj, 31u - bb), wb(wi, a));
      sx = x + sc;
    }
  }
  acc_add(acc, sx);
#pragma unroll 1
  for (uint32_t r = 0; r < 8u / kFarNG; r++) {
    float g[kFarNG];
    float ln[kFarNG];
    uint32_t cls[kFarNG];
#pragma unroll
    for (int u = 0; u < kFarNG; u++) {
      const uint32_t p = t + (kFarNG * r + static_cast<uint32_t>(u)) * 64u;
      uint32_t a = 0, bb = 0;
      const bool ok = probe_slot(p, a, bb) && !Special<CONTRA>::is(a, bb) && a + bb + 3u <= d;
      cls[u] = slot_class(a, bb);
      ln[u] = ok ? tlen[p] : kNegInf;
      g[u] = ok ? qx[cls[u] * msz + static_cast<size_t>(i + 1u + a) * ld + (j - 1u - bb)] : 0.f;
    }
    float xv[kFarNG];
#pragma unroll
    for (int u = 0; u < kFarNG; u++) xv[u] = (g[u] + ln[u]) + pick(cs4, cls[u]);
#pragma unroll
    for (int u = 0; u < kFarNG; u += 4) acc_add4(acc, xv[u], xv[u + 1], xv[u + 2], xv[u + 3]);
  }
}

// enclosing 2-loops of the finished pair (i,j) (574-593 / 681-700): (k,l) = (i-1-a, j+1+b)
// closes, (i,j) is enclosed; wi = bases i-31 .. i, wj = bases j .. j+31
template <bool CONTRA, int TPC>
__device__ __forceinline__ void outer_block(const TreeBatch& b, const TSeq& q, Acc& acc, uint32_t i,
                                            uint32_t j, uint32_t t, float qb, const float4& in4,
                                            uint64_t wi, uint64_t wj) {
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 1) return;
#endif
  const uint32_t n = q.n, ld = q.ld;
  const float* __restrict__ px = q.m[T_X4];  // four planes, one per class
  const size_t msz = static_cast<size_t>(q.m[1] - q.m[0]);
  const float* __restrict__ tlen = &b.tabs->len[CONTRA ? 1 : 0][0];
  constexpr int NP = (512 + TPC - 1) / TPC;
  float g[NP];
  float ln[NP];
  uint32_t cls[NP];
#pragma unroll
  for (int u = 0; u < NP; u++) {
    const uint32_t p = t + static_cast<uint32_t>(u) * TPC;
    uint32_t a = 0, bb = 0;
    const bool ok = p < 512u && probe_slot(p, a, bb) && !Special<CONTRA>::is(a, bb) && a < i && j + 1u + bb < n;
    cls[u] = slot_class(a, bb);
    ln[u] = ok ? tlen[p] : kNegInf;
    g[u] = ok ? px[cls[u] * msz + static_cast<size_t>(i - 1u - a) * ld + (j + 1u + bb)] : 0.f;
  }
  float sx = kNegInf;
  if (t < Special<CONTRA>::N) {
    uint32_t a, bb;
    Special<CONTRA>::slot(t, a, bb);
    if (a < i && j + 1u + bb < n) {
      // closing pair at window positions 30-a / 1+bb, its inner neighbours at 31-a, 32-a / bb, bb-1
      const uint32_t k = i - 1u - a, l = j + 1u + bb;
      const float x = q.m[T_QB][static_cast<size_t>(k) * ld + l];
      const float pkl = q.out[tri_off(n, l - k) + k];
      const float sc = TModel<CONTRA>::twoloop(b, a, bb, wb(wi, 30u - a), wb(wj, 1u + bb), wb(wi, 31u - a),
                                               wb(wi, a >= 1u ? 32u - a : 31u), wb(wj, bb),
                                               wb(wj, bb >= 1u ? bb - 1u : 0u), wb(wi, 31), wb(wj, 0),
                                               wb(wj, 1), wb(wi, 30));
      if (x > kNegInf) sx = ((pkl + qb) - x) + sc;
    }
  }
  float xv[NP], mx = sx;  // (absent pair: its slot holds -inf)
#pragma unroll
  for (int u = 0; u < NP; u++) {
    xv[u] = ((g[u] + qb) + ln[u]) + pick(in4, cls[u]);
    mx = vmaxf(mx, xv[u]);
  }
  mx = vmaxf(mx, kEmpty);
  float sm = ex2((sx - mx) * kL2E);
#pragma unroll
  for (int u = 0; u < NP; u++) sm += ex2((xv[u] - mx) * kL2E);
  acc_merge(acc, Acc{mx, sm});
}

// far part of the same, as pair_far
template <bool CONTRA>
__device__ __forceinline__ void outer_far(const TreeBatch& b, const TSeq& q, Acc& acc, uint32_t i, uint32_t j,
                                          uint32_t t, float qb, const float4& in4, uint64_t wi, uint64_t wj) {
  const uint32_t n = q.n, ld = q.ld;
  const float* __restrict__ px = q.m[T_X4];  // four planes, one per class
  const size_t msz = static_cast<size_t>(q.m[1] - q.m[0]);
  const float* __restrict__ tlen = &b.tabs->len[CONTRA ? 1 : 0][0];
  float sx = kNegInf;
  if (t >= kNear && t < Special<CONTRA>::N) {
    uint32_t a, bb;
    Special<CONTRA>::slot(t, a, bb);
    if (a < i && j + 1u + bb < n) {
      const uint32_t k = i - 1u - a, l = j + 1u + bb;
      const float x = q.m[T_QB][static_cast<size_t>(k) * ld + l];
      const float pkl = q.out[tri_off(n, l - k) + k];
      const float sc = TModel<CONTRA>::twoloop(b, a, bb, wb(wi, 30u - a), wb(wj, 1u + bb), wb(wi, 31u - a),
                                               wb(wi, a >= 1u ? 32u - a : 31u), wb(wj, bb),
                                               wb(wj, bb >= 1u ? bb - 1u : 0u), wb(wi, 31), wb(wj, 0),
                                               wb(wj, 1), wb(wi, 30));
      if (x > kNegInf) sx = ((pkl + qb) - x) + sc;
    }
  }
  acc_add(acc, sx);
#pragma unroll 1
  for (uint32_t r = 0; r < 8u / kFarNG; r++) {
    float g[kFarNG];
    float ln[kFarNG];
    uint32_t cls[kFarNG];
#pragma unroll
    for (int u = 0; u < kFarNG; u++) {
      const uint32_t p = t + (kFarNG * r + static_cast<uint32_t>(u)) * 64u;
      uint32_t a = 0, bb = 0;
      const bool ok = probe_slot(p, a, bb) && !Special<CONTRA>::is(a, bb) && a < i && j + 1u + bb < n;
      cls[u] = slot_class(a, bb);
      ln[u] = ok ? tlen[p] : kNegInf;
      g[u] = ok ? px[cls[u] * msz + static_cast<size_t>(i - 1u - a) * ld + (j + 1u + bb)] : 0.f;
    }
    float xv[kFarNG];
#pragma unroll
    for (int u = 0; u < kFarNG; u++) xv[u] = ((g[u] + qb) + ln[u]) + pick(in4, cls[u]);
#pragma unroll
    for (int u = 0; u < kFarNG; u += 4) acc_add4(acc, xv[u], xv[u + 1], xv[u + 2], xv[u + 3]);
  }
}

// ----------------------------------------------------------------------------
// The uniform operands of a cell pair in ONE vector gather.  A launch of the banded sweep needs ~60
// scalars per cell pair (statics, the neighbours' sums, ring entries).  As scalar loads behind
// `cond ? sload(..) : default` the compiler emitted them as ~14 conditional s_load / s_waitcnt pairs
// in SEQUENCE — 4.3 us of a launch's ~11.7 (tree_debug 8 against 4, round 4) — because a scalar
// load cannot be predicated.  Here lane x of the wave loads operand x: a descriptor per lane
// (constant table: which matrix, which neighbour cell, which condition, which default), one
// `global_load_dword` for all of them, `v_readlane` hands each to the scalar code.  One round trip,
// and the near-slot gathers and the product streams are issued right behind it.
//   kind 0: row-major matrix, cell (i + a, j + b)          1: float4 per cell, row-major, component c
//        2: column-major matrix, cell (i + b, j + a)        3: row-major, cell (i + a, i + b)
//        4: far ring, diagonal d + a, row i + b, comp c     5..7: mid ring of product kind - 5, diagonal d + a, row i + b, comp c
//        8: sums_external's prefix vector zp[a ? n : i + b]  9: its suffix vector zs[j + b]
//   cond bits: 1 has1, 2 d >= 1, 4 d >= 2, 8 CONTRAfold only, 16 banded (thr != 0), 32 i + b >= 0, 64 j + 1 < n,
//              128 j + 2 < n, 256 not has1
//   dflt: 0 -inf, 1 zero, 2 kEmpty (the max of an empty accumulator)
constexpr uint32_t opd(uint32_t kind, uint32_t mat, int a, int bb, uint32_t comp, uint32_t cond, uint32_t dflt) {
  return kind | (mat << 4) | (static_cast<uint32_t>(a + 2) << 9) | (static_cast<uint32_t>(bb + 4) << 12) | (comp << 16) |
         (cond << 18) | (dflt << 27);
}
constexpr uint32_t kOpNone = 15u;
struct OpCtx {
  float* hbase;     // first matrix of the sequence
  uint64_t hmsz;
  const float2* far;
  const float2* mid;
  const float* zp;
  const float* zs;
  uint32_t i, j, d, n, ld, vec, ring;
  bool has1, contra, banded;
};
__device__ __forceinline__ float gather_operand(uint32_t ds, const OpCtx& c) {
  const uint32_t kind = ds & 15u, mat = (ds >> 4) & 31u, comp = (ds >> 16) & 3u, cond = (ds >> 18) & 511u,
                 dflt = (ds >> 27) & 3u;
  const int a = static_cast<int>((ds >> 9) & 7u) - 2, bb = static_cast<int>((ds >> 12) & 15u) - 4;
  bool ok = kind != kOpNone;
  ok = ok && (!(cond & 1u) || c.has1) && (!(cond & 2u) || c.d >= 1u) && (!(cond & 4u) || c.d >= 2u) &&
       (!(cond & 8u) || c.contra) && (!(cond & 16u) || c.banded) && (!(cond & 32u) || static_cast<int>(c.i) + bb >= 0) &&
       (!(cond & 64u) || c.j + 1u < c.n) && (!(cond & 128u) || c.j + 2u < c.n) && (!(cond & 256u) || !c.has1);
  const float* p = c.hbase;
  const int64_t ia = static_cast<int64_t>(c.i) + a, ib = static_cast<int64_t>(c.i) + bb;
  if (kind <= 3u) {
    size_t off;
    if (kind == 2u) off = static_cast<size_t>(static_cast<int64_t>(c.j) + a) * c.ld + static_cast<size_t>(ib);
    else if (kind == 3u) off = static_cast<size_t>(ia) * c.ld + static_cast<size_t>(ib);
    else off = static_cast<size_t>(ia) * c.ld + static_cast<size_t>(static_cast<int64_t>(c.j) + bb);
    if (kind == 1u) off = 4u * off + comp;
    p = c.hbase + static_cast<size_t>(mat) * c.hmsz + off;
  } else if (kind == 4u) {
    p = reinterpret_cast<const float*>(c.far + static_cast<size_t>((c.d + static_cast<uint32_t>(a)) & 3u) * c.vec +
                                       static_cast<size_t>(ib)) + comp;
  } else if (kind <= 7u) {
    const uint32_t ring = max(c.ring, 1u);
    p = reinterpret_cast<const float*>(c.mid + (static_cast<size_t>(kind - 5u) * c.ring + (c.d + static_cast<uint32_t>(a)) % ring) * c.vec +
                                       static_cast<size_t>(ib)) + comp;
  } else if (kind == 8u) {
    p = c.zp + (a ? static_cast<size_t>(c.n) : static_cast<size_t>(ib));
  } else {
    p = c.zs + static_cast<size_t>(static_cast<int64_t>(c.j) + bb);
  }
  const float v = ok ? *p : 0.f;
  return ok ? v : (dflt == 0u ? kNegInf : (dflt == 1u ? 0.f : kEmpty));
}
__device__ __forceinline__ float* hbase_of(const TSeq& q) { return q.m[0]; }
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(v)), lane)));
}
// inside sweep, cells (i,j), (i,j+1), neighbour (i+1,j+1)
enum InOp : int {
  IO_MBC0, IO_MBC1, IO_MBCN, IO_HP0, IO_HP1, IO_HPN, IO_ACCS0, IO_ACCS1, IO_ACCSN,
  IO_CS0, IO_IN0 = IO_CS0 + 4, IO_CS1 = IO_IN0 + 4, IO_IN1 = IO_CS1 + 4, IO_CSN = IO_IN1 + 4,
  IO_QM0 = IO_CSN + 4, IO_QM1, IO_QMN, IO_ZRE_P, IO_ZRM_P, IO_U_N0, IO_ZRE_PN, IO_ZRM_PN, IO_U_NN, IO_Q1II_A, IO_Q1II_B,
  IO_FAR0, IO_FARN = IO_FAR0 + 2, IO_FAR1 = IO_FARN + 2, IO_MID0 = IO_FAR1 + 2, IO_MID1 = IO_MID0 + 2, IO_COUNT = IO_MID1 + 2
};
static_assert(IO_COUNT <= 64, "one lane per operand");
__constant__ uint32_t kInOps[64] = {
    opd(0, T_MBC, 0, 0, 0, 0, 0), opd(0, T_MBC, 0, 1, 0, 1, 0), opd(0, T_MBC, 1, 1, 0, 1, 0),
    opd(0, T_HP, 0, 0, 0, 0, 0), opd(0, T_HP, 0, 1, 0, 1, 0), opd(0, T_HP, 1, 1, 0, 1, 0),
    opd(0, T_ACCS, 0, 0, 0, 0, 1), opd(0, T_ACCS, 0, 1, 0, 1, 1), opd(0, T_ACCS, 1, 1, 0, 1, 1),
    opd(1, T_CS4, 0, 0, 0, 0, 1), opd(1, T_CS4, 0, 0, 1, 0, 1), opd(1, T_CS4, 0, 0, 2, 0, 1), opd(1, T_CS4, 0, 0, 3, 0, 1),
    opd(1, T_IN4, 0, 0, 0, 0, 1), opd(1, T_IN4, 0, 0, 1, 0, 1), opd(1, T_IN4, 0, 0, 2, 0, 1), opd(1, T_IN4, 0, 0, 3, 0, 1),
    opd(1, T_CS4, 0, 1, 0, 1, 1), opd(1, T_CS4, 0, 1, 1, 1, 1), opd(1, T_CS4, 0, 1, 2, 1, 1), opd(1, T_CS4, 0, 1, 3, 1, 1),
    opd(1, T_IN4, 0, 1, 0, 1, 1), opd(1, T_IN4, 0, 1, 1, 1, 1), opd(1, T_IN4, 0, 1, 2, 1, 1), opd(1, T_IN4, 0, 1, 3, 1, 1),
    opd(1, T_CS4, 1, 1, 0, 1, 1), opd(1, T_CS4, 1, 1, 1, 1, 1), opd(1, T_CS4, 1, 1, 2, 1, 1), opd(1, T_CS4, 1, 1, 3, 1, 1),
    opd(0, T_QM, 1, -1, 0, 4, 0), opd(0, T_QM, 1, 0, 0, 1 | 2, 0), opd(0, T_QM, 2, 0, 0, 1 | 4, 0),
    opd(2, T_ZRE, -1, 0, 0, 0, 0), opd(2, T_ZRM, -1, 0, 0, 8, 0), opd(2, T_U, 0, 1, 0, 0, 0),
    opd(2, T_ZRE, 0, 1, 0, 1, 0), opd(2, T_ZRM, 0, 1, 0, 1 | 8, 0), opd(2, T_U, 1, 2, 0, 1, 0),
    opd(3, T_Q1R, 0, 0, 0, 1 | 4, 0), opd(2, T_ZRM, 0, 1, 0, 1 | 4, 0),
    opd(4, 0, 0, 0, 0, 0, 2), opd(4, 0, 0, 0, 1, 0, 1), opd(4, 0, 0, 1, 0, 1, 2), opd(4, 0, 0, 1, 1, 1, 1),
    opd(4, 0, 1, 0, 0, 1, 2), opd(4, 0, 1, 0, 1, 1, 1),
    opd(5, 0, 0, 0, 0, 16, 2), opd(5, 0, 0, 0, 1, 16, 1), opd(5, 0, 1, 0, 0, 1 | 16, 2), opd(5, 0, 1, 0, 1, 1 | 16, 1),
    kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone};
// outside sweep, cells (i,j+1) and (i,j) (and probs_multibranch of (i-1,j) from the ring)
enum OutOp : int {
  OO_QB0, OO_QB1, OO_PM2_A, OO_PM2_B, OO_W_A, OO_W_B, OO_SP1, OO_SP2, OO_ZTOT, OO_ZPI, OO_ZSJ0, OO_ZSJ1,
  OO_QA0, OO_MBC0, OO_QA1, OO_MBC1, OO_CS0, OO_CS1 = OO_CS0 + 4, OO_IN0 = OO_CS1 + 4, OO_IN1 = OO_IN0 + 4,
  OO_FAR0 = OO_IN1 + 4, OO_FAR1 = OO_FAR0 + 2, OO_P0 = OO_FAR1 + 2, OO_PN = OO_P0 + 2, OO_P1 = OO_PN + 2,
  OO_E0 = OO_P1 + 2, OO_E1 = OO_E0 + 2, OO_COUNT = OO_E1 + 2
};
static_assert(OO_COUNT <= 64, "one lane per operand");
__constant__ uint32_t kOutOps[64] = {
    opd(0, T_QB, 0, 0, 0, 0, 0), opd(0, T_QB, 0, 1, 0, 1, 0),
    // Pm2 / W of the cell right of the group's upper cell: (i, j+2) with a second cell, (i, j+1) without
    opd(0, T_QM, 0, 2, 0, 1 | 128, 0), opd(0, T_QM, 0, 1, 0, 256 | 64, 0),
    opd(0, T_ZRE, 0, 2, 0, 1 | 128, 0), opd(0, T_ZRE, 0, 1, 0, 256 | 64, 0),
    opd(2, T_U, 1, -1, 0, 1 | 32, 0), opd(2, T_U, 0, -2, 0, 32, 0),
    opd(8, 0, 1, 0, 0, 0, 1), opd(8, 0, 0, 0, 0, 0, 1), opd(9, 0, 0, 1, 0, 0, 1), opd(9, 0, 0, 2, 0, 1, 1),
    opd(0, T_QA, 0, 0, 0, 0, 0), opd(0, T_MBC, 0, 0, 0, 0, 0), opd(0, T_QA, 0, 1, 0, 1, 0), opd(0, T_MBC, 0, 1, 0, 1, 0),
    opd(1, T_CS4, 0, 0, 0, 0, 1), opd(1, T_CS4, 0, 0, 1, 0, 1), opd(1, T_CS4, 0, 0, 2, 0, 1), opd(1, T_CS4, 0, 0, 3, 0, 1),
    opd(1, T_CS4, 0, 1, 0, 1, 1), opd(1, T_CS4, 0, 1, 1, 1, 1), opd(1, T_CS4, 0, 1, 2, 1, 1), opd(1, T_CS4, 0, 1, 3, 1, 1),
    opd(1, T_IN4, 0, 0, 0, 0, 1), opd(1, T_IN4, 0, 0, 1, 0, 1), opd(1, T_IN4, 0, 0, 2, 0, 1), opd(1, T_IN4, 0, 0, 3, 0, 1),
    opd(1, T_IN4, 0, 1, 0, 1, 1), opd(1, T_IN4, 0, 1, 1, 1, 1), opd(1, T_IN4, 0, 1, 2, 1, 1), opd(1, T_IN4, 0, 1, 3, 1, 1),
    opd(4, 0, 0, 0, 0, 0, 2), opd(4, 0, 0, 0, 1, 0, 1), opd(4, 0, 1, 0, 0, 1, 2), opd(4, 0, 1, 0, 1, 1, 1),
    opd(6, 0, 0, 0, 0, 16, 2), opd(6, 0, 0, 0, 1, 16, 1),
    opd(6, 0, 1, -1, 0, 16 | 32 | 64, 2), opd(6, 0, 1, -1, 1, 16 | 32 | 64, 1),
    opd(6, 0, 1, 0, 0, 16 | 1, 2), opd(6, 0, 1, 0, 1, 16 | 1, 1),
    opd(7, 0, 0, 0, 0, 16, 2), opd(7, 0, 0, 0, 1, 16, 1), opd(7, 0, 1, 0, 0, 16 | 1, 2), opd(7, 0, 1, 0, 1, 16 | 1, 1),
    kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone, kOpNone,
    kOpNone, kOpNone, kOpNone, kOpNone};

// ----------------------------------------------------------------------------
// Two diagonals per launch.  The sweep's cost is its NUMBER of dependent launches and the
// dependent round trips inside each, so one group of TPC threads takes the cells (i, j) and
// (i, j+1) of diagonals d and d+1 (`single`: the cell (i,j) alone).  What diagonal d+1 needs of
// diagonal d:
//   inside : Zr(i,j) (own cell) and U(i+1,j+1), one step from U(i+2,j+1) given the
//            closing-pair block of the neighbour (i+1,j+1), which is evaluated a second time here
//            (sums_multibranch(i,j+1) itself does not: its k = i+1 term carries Q1(i,i) = -inf);
//   outside: W(i,j+1), Pm2(i,j+1) (own cell) and probs_multibranch(i-1,j) for the column
//            prefix, a product that shares its Q1 row with the cell's own and is evaluated here
//            as a third stream (the k = j+1 term of Pm(i,j) carries Q1(j+1,j) = -inf, and the
//            k = i-1 term of L_e Q1(i,i-1) = -inf: neither needs the neighbour).
// TPC: 64 (four cells per workgroup, no barrier), 256 or 1024 threads per cell; the host picks
// by the length of the cell's sums.  Scalars of a cell are computed by all of its threads alike
// (wave-uniform: scalar unit); its first lane stores.

// (+)_k over idx in [0, len1): a = A[idx]; acc0 += a + B0[idx] (idx < len0); acc1 += a + B1[idx]
template <int TPC>
__device__ __forceinline__ void acc_product_2b(Acc& acc0, Acc& acc1, const float* __restrict__ A,
                                               const float* __restrict__ B0,
                                               const float* __restrict__ B1, uint32_t len0,
                                               uint32_t len1, uint32_t t) {
  for (uint32_t k = t; k < len1; k += 4u * TPC) {
    float x0[4], x1[4], va[4], vp[4], vr[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t ku = k + static_cast<uint32_t>(u) * TPC;
      const uint32_t k1 = ku < len1 ? ku : 0u, k0 = ku < len0 ? ku : 0u;
      va[u] = A[k1], vp[u] = B0[k0], vr[u] = B1[k1];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t ku = k + static_cast<uint32_t>(u) * TPC;
      x0[u] = ku < len0 ? va[u] + vp[u] : kNegInf;
      x1[u] = ku < len1 ? va[u] + vr[u] : kNegInf;
    }
    acc_add4(acc0, x0[0], x0[1], x0[2], x0[3]);
    acc_add4(acc1, x1[0], x1[1], x1[2], x1[3]);
  }
}

// The same two-cell product over the EDGE of a banded cell (see k_tree_mid): e in [0, tot),
// idx = e (e < L) or e + jump; the first cell's stream ends at lim0l (left piece) / lim0r (right)
template <int TPC>
__device__ __forceinline__ void acc_product_2b_split(Acc& acc0, Acc& acc1, const float* __restrict__ A,
                                                     const float* __restrict__ B0,
                                                     const float* __restrict__ B1, uint32_t L, uint32_t tot,
                                                     uint32_t jump, uint32_t lim0l, uint32_t lim0r,
                                                     uint32_t t) {
  for (uint32_t k = t; k < tot; k += 4u * TPC) {
    float x0[4], x1[4], va[4], vp[4], vr[4];
    bool ok0[4], ok1[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t e = k + static_cast<uint32_t>(u) * TPC;
      const bool left = e < L;
      const uint32_t idx = left ? e : e + jump;
      ok1[u] = e < tot;
      ok0[u] = ok1[u] && idx < (left ? lim0l : lim0r);
      const uint32_t i1 = ok1[u] ? idx : 0u, i0 = ok0[u] ? idx : 0u;
      va[u] = A[i1], vp[u] = B0[i0], vr[u] = B1[i1];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; u++) {
      x0[u] = ok0[u] ? va[u] + vp[u] : kNegInf;
      x1[u] = ok1[u] ? va[u] + vr[u] : kNegInf;
    }
    acc_add4(acc0, x0[0], x0[1], x0[2], x0[3]);
    acc_add4(acc1, x1[0], x1[1], x1[2], x1[3]);
  }
}
template <int TPC>
__device__ __forceinline__ void acc_product_split(Acc& a, const float* __restrict__ A,
                                                  const float* __restrict__ B, uint32_t L, uint32_t tot,
                                                  uint32_t jump, uint32_t t) {
  for (uint32_t k = t; k < tot; k += 8u * TPC) {
    float x[8], va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t e = k + static_cast<uint32_t>(u) * TPC;
      const uint32_t idx = e < tot ? (e < L ? e : e + jump) : 0u;
      va[u] = A[idx], vb[u] = B[idx];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; u++) x[u] = (k + static_cast<uint32_t>(u) * TPC < tot) ? va[u] + vb[u] : kNegInf;
    acc_add4(a, x[0], x[1], x[2], x[3]);
    acc_add4(a, x[4], x[5], x[6], x[7]);
  }
}
// acc_product_2b whose second stream starts at idx = lo1
template <int TPC>
__device__ __forceinline__ void acc_product_2b_lo(Acc& acc0, Acc& acc1, const float* __restrict__ A,
                                                  const float* __restrict__ B0,
                                                  const float* __restrict__ B1, uint32_t len0,
                                                  uint32_t len1, uint32_t lo1, uint32_t t) {
  for (uint32_t k = t; k < len1; k += 4u * TPC) {
    float x0[4], x1[4], va[4], vp[4], vq[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t ku = k + static_cast<uint32_t>(u) * TPC;
      const bool v1 = ku < len1, v0 = ku < len0, vr = v1 && ku >= lo1;
      va[u] = A[v1 ? ku : 0u], vp[u] = B0[v0 ? ku : 0u], vq[u] = B1[vr ? ku : lo1 < len1 ? lo1 : 0u];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t ku = k + static_cast<uint32_t>(u) * TPC;
      x0[u] = ku < len0 ? va[u] + vp[u] : kNegInf;
      x1[u] = (ku < len1 && ku >= lo1) ? va[u] + vq[u] : kNegInf;
    }
    acc_add4(acc0, x0[0], x0[1], x0[2], x0[3]);
    acc_add4(acc1, x1[0], x1[1], x1[2], x1[3]);
  }
}
// Far parts of the closing-pair blocks (inside) / enclosing 2-loops (outside) of a diagonal's
// cells, written one launch ahead: ring of four diagonals, {max, sum} per cell
__device__ __forceinline__ Acc load_far(const TSeq& q, uint32_t d, uint32_t i) {
  const float2 v = sload2(q.far + static_cast<size_t>(d & 3u) * q.vec + i);
  return Acc{v.x, v.y};
}
__device__ __forceinline__ Acc load_mid(const TSeq& q, uint32_t ring, uint32_t prod, uint32_t d, uint32_t i) {
  const float2 v = sload2(q.mid + (static_cast<size_t>(prod) * ring + d % ring) * q.vec + i);
  return Acc{v.x, v.y};
}

// `thr` (inside): banded cells.  The terms of sums_multibranch(i,j) whose two operands both span
// less than thr were summed by k_tree_mid before this band started (they need nothing of the
// last two bands); the launch adds the rest (at most 2 (d - thr) terms: the EDGE) and merges.
// UF: the far parts of this launch's blocks come from q.far (TPC == 64 only; a variant of its own
// so that neither form carries the other's registers)
template <bool CONTRA, int TPC, bool UF>
__global__ void __launch_bounds__(TPC < 256 ? 256 : TPC) k_tree_inside2(float* hbase, uint64_t hmsz, uint32_t hn, uint32_t hld,
                                                                        uint32_t d, uint32_t thr, uint32_t use_one,
                                                                        TreeBatch b, int single, Ahead ah) {
  static_assert(!UF || TPC == 64, "far parts from the ring: one wave per cell pair");
  constexpr int BLOCK = TPC < 256 ? 256 : TPC;
  constexpr int NA = 9;
  __shared__ float red[BLOCK / 64][NA][2];
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 4) return;
#endif
  const TSeq q = load_tseq_hot(b, blockIdx.y, hbase, hmsz, hn, hld, use_one);
  const uint32_t n = q.n, ld = q.ld;
  const uint32_t bx = role_block(ah);
  if (bx >= ah.main_blocks) {
    // ---- the NEXT launch's closing-pair blocks, far part (see Ahead): one wave per row i takes
    // the cells (i, i+nd0) and (i, i+nd0+1); nothing here depends on this launch's cells
#ifdef RNAMC_DEBUG_KNOBS
    if (b.debug & 1) return;
#endif
    // (flags bit 1: one wave per CELL instead of one per row — the launch has room for them)
    __builtin_amdgcn_s_setprio(1);
    const bool split = (ah.flags & 2u) != 0u;
    const uint32_t wid = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
        static_cast<int>((bx - ah.main_blocks) * (BLOCK / 64) + threadIdx.x / 64)));
    const uint32_t i = split ? wid >> 1 : wid;
    const bool doA = !split || (wid & 1u) == 0u, doB = !split || (wid & 1u) != 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const size_t row_i = static_cast<size_t>(i) * ld;
    // (both cells' uniform operands in one round trip; a cell past the row's end reads the pad)
    const uint32_t j0 = i + ah.nd0;
    if (j0 >= n) return;
    const uint64_t wi = load_win64(q.pk, static_cast<int>(i));
    const bool two = doB && ah.nd_count > 1u && j0 + 1u < n;
    const float mbcA = doA ? sload(q.m[T_MBC] + row_i + j0) : kNegInf;
    const float mbcB = two ? sload(q.m[T_MBC] + row_i + j0 + 1u) : kNegInf;
    const float4 csA = sload4(reinterpret_cast<const float4*>(q.m[T_CS4]) + row_i + j0);
    const float4 csB = sload4(reinterpret_cast<const float4*>(q.m[T_CS4]) + row_i + j0 + (two ? 1u : 0u));
    const uint64_t wjA = load_win64(q.pk, static_cast<int>(j0) - 31);
    const uint64_t wjB = load_win64(q.pk, static_cast<int>(j0) - 30);
    Acc fA = acc_empty(), fB = acc_empty();
    if (mbcA > kNegInf) pair_far<CONTRA>(b, q, fA, i, j0, lane, csA, wi, wjA);
    if (mbcB > kNegInf) pair_far<CONTRA>(b, q, fB, i, j0 + 1u, lane, csB, wi, wjB);
    if (mbcA > kNegInf) fA = wave_reduce(fA);
    if (mbcB > kNegInf) fB = wave_reduce(fB);
    if (lane == 0u) {
      if (doA) q.far[static_cast<size_t>(ah.nd0 & 3u) * q.vec + i] = make_float2(fA.m, fA.s);
      if (two) q.far[static_cast<size_t>((ah.nd0 + 1u) & 3u) * q.vec + i] = make_float2(fB.m, fB.s);
    }
    return;
  }
  // (wave-uniform: TPC is a multiple of 64)
  const uint32_t i = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
      static_cast<int>(bx * (BLOCK / TPC) + threadIdx.x / TPC)));
  if (i + d >= n) return;
  // the launch's chain goes first at the issue ports: the mid-field kernel beside it (priority 0)
  // and the ahead waves (1) take what it leaves
  __builtin_amdgcn_s_setprio(3);
  constexpr bool uf = UF;
  const uint32_t j = i + d, j1 = j + 1u;
  const bool has1 = !single && j1 < n;  // cells (i, j+1) and (i+1, j+1) are this launch's too
  const uint32_t t = threadIdx.x % TPC;
  const size_t row_i = static_cast<size_t>(i) * ld, col_j = static_cast<size_t>(j) * ld;
  const float4* __restrict__ cs4m = reinterpret_cast<const float4*>(q.m[T_CS4]);
  const float4* __restrict__ in4m = reinterpret_cast<const float4*>(q.m[T_IN4]);

  // Roles inside a cell's group of TPC > 64 threads: its first three waves take one
  // closing-pair block each (eight probe slots per lane) and join the products afterwards, the
  // others stream products from the start; the first wave alone fetches the operands of the
  // scalar recurrences and runs the epilogue.  (TPC == 64: the one wave does everything.)
  const uint32_t wv = TPC == 64 ? 0u : static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(t >> 6)));
  const bool w0 = wv == 0u;
  // (one wave: all three blocks; two waves: (i,j) and (i,j+1) in the first, the neighbour's in the second)
  // (far parts taken from q.far: what is left of the three blocks is the first wave's)
  const bool do0 = w0, don = uf ? w0 : (TPC == 64 || wv == 1u),
             do1 = uf ? w0 : (TPC == 64 || wv == (TPC == 128 ? 0u : 2u));
  const uint32_t lane = t & 63u;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  Acc acc[NA];
#pragma unroll
  for (int x = 0; x < NA; x++) acc[x] = acc_empty();
  // (banded sweeps: sums_external's row 0 and column n-1 are k_tree_ext's, a band behind)
  const bool ext_side = b.ring != 0u;
  const bool zs0 = !ext_side && j == n - 1, zs1 = !ext_side && has1 && j1 == n - 1;  // which cell sits in column n-1
  const bool row0 = !ext_side && i == 0;
  // [3] [4] sums_multibranch of (i,j) and (i,j+1): k = i+1 .. j-1 | j, Q1(i,k-1) + Zr_mb(k,j | j+1)
  // (the k = i+1 term of the second reads a cell of this launch: left out, it carries Q1(i,i) = -inf)
  auto products = [&]() {
#ifdef RNAMC_DEBUG_KNOBS
    if (b.debug & 2) return;
#endif
    if (thr != 0u) {
      // idx: Q1(i, i+1+idx) [span idx+1] + Zr_mb(i+2+idx, j | j+1) [span d-2-idx | d-1-idx]
      if (has1)
        acc_product_2b_split<TPC>(acc[3], acc[4], q.m[T_Q1R] + row_i + i + 1, q.m[T_ZRM] + col_j + i + 2,
                                  q.m[T_ZRM] + col_j + ld + i + 2, d - thr, 2u * (d - thr),
                                  (thr - 1u) - (d - thr), d - 1u - thr, d - 2u, t);
      else  // idx': Q1(i, i+idx') [span idx'] + Zr_mb(i+1+idx', j) [span d-1-idx']
        acc_product_split<TPC>(acc[3], q.m[T_Q1R] + row_i + i, q.m[T_ZRM] + col_j + i + 1, d - thr,
                               (d - thr) + (d - 1u - thr), thr - (d - thr), t);
    } else if (has1) {
      if (d >= 2)
        acc_product_2b<TPC>(acc[3], acc[4], q.m[T_Q1R] + row_i + i + 1, q.m[T_ZRM] + col_j + i + 2,
                            q.m[T_ZRM] + col_j + ld + i + 2, d - 2, d - 1, t);
      // (k = i+1 of the first, left out to keep both streams on the same k: q1_ii, added below)
    } else if (d >= 2) {
      acc_product<TPC>(acc[3], q.m[T_Q1R] + row_i + i, q.m[T_ZRM] + col_j + i + 1, d - 1, t);
    }
    // [5] [6] Z(0,j), Z(0,j+1): k >= 1 | 2, Zr_ext(k,.) + Z(0,k-1)
    if (row0) {
      if (j >= 1) acc_product<TPC>(acc[5], q.m[T_ZRE] + col_j + 1, q.zp + 1, j, t);
      if (has1 && j1 >= 2) acc_product<TPC>(acc[6], q.m[T_ZRE] + col_j + ld + 2, q.zp + 2, j1 - 1, t);
    }
    // [7] column n-1 cell of this group: l = i+1 .. (zs0: n-2 | zs1: n-3), Qa(i,l) + Z(l+1,n-1)
    //     (zs1: l = j = n-2 is this launch's own cell (i,j))
    if ((zs0 || zs1) && d >= 1) acc_product<TPC>(acc[7], q.m[T_QA] + row_i + i + 1, q.zs + i + 2, d - 1, t);
    // [8] zs1: the neighbour (i+1,n-1)'s own sum, l = i+2 .. n-2
    if (zs1 && d >= 1)
      acc_product<TPC>(acc[8], q.m[T_QA] + row_i + ld + i + 2, q.zs + i + 3, d - 1, t);
  };
  // every uniform operand of the cell pair: one vector gather in the banded sweep's launches (uf:
  // gather_operand above), scalar loads in the other variants
  float mbc0 = kNegInf, mbc1 = kNegInf, mbcn = kNegInf, hp0 = kNegInf, hp1 = kNegInf, hpn = kNegInf;
  float accs0 = 0.f, accs1 = 0.f, accsn = 0.f;
  float4 cs0 = zero4, in0 = zero4, cs1 = zero4, in1 = zero4, csn = zero4;
  float qm0 = kNegInf, qm1 = kNegInf, qmn = kNegInf;
  float zr_e_prev = kNegInf, zr_m_prev = kNegInf, zr_e_prevn = kNegInf, zr_m_prevn = kNegInf;
  float u_next0 = kNegInf, u_nextn = kNegInf;  // U(i+1, j), U(i+2, j+1)
  float zs_a = 0.f, zs_last = 0.f, zp1 = 0.f, q1_ii = kNegInf;
  Acc far0 = acc_empty(), farn = acc_empty(), far1 = acc_empty(), mid0 = acc_empty(), mid1 = acc_empty();
  float nqb = kNegInf, nsc = 0.f;  // (uf) lane 3c + s: sums_close of the pair near slot s of cell c encloses, its score
  bool ngeo = false;
  if constexpr (uf) {
    const OpCtx oc{hbase_of(q), static_cast<uint64_t>(q.m[1] - q.m[0]), q.far, q.mid, q.zp, q.zs, i, j, d, n, ld, q.vec, b.ring,
                   has1, CONTRA, thr != 0u};
    const float ov = gather_operand(kInOps[lane], oc);
    // (issued right behind the gather, before anything waits for it: the near slots' sums_close and
    // scores — slot s of cell c in lane 3c + s — are gathers of their own; whether the cell may pair
    // at all is applied to the VALUE below)
    {
      const uint32_t c = lane / 3u, sl = lane - 3u * c;
      const uint32_t ci = c == 1u ? i + 1u : i, cj = c == 0u ? j : j1;
      uint32_t sa, sb;
      Special<CONTRA>::slot(sl < kNear ? sl : 0u, sa, sb);
      ngeo = lane < 9u && (c == 0u || has1) && sa + sb + 3u <= cj - ci;
      if (ngeo) {
        nqb = q.m[T_QB][static_cast<size_t>(ci + 1u + sa) * ld + (cj - 1u - sb)];
        nsc = q.m[T_NEAR4][4u * (static_cast<size_t>(ci) * ld + cj) + sl];
      }
    }
    // (the product streams need none of the operands: their loads go out behind the gather, and their
    // sums run while it is in flight — one round trip for both)
    products();
    auto OP = [&](int x) { return lane_value(ov, x); };
    mbc0 = OP(IO_MBC0), mbc1 = OP(IO_MBC1), mbcn = OP(IO_MBCN);
    hp0 = OP(IO_HP0), hp1 = OP(IO_HP1), hpn = OP(IO_HPN);
    accs0 = OP(IO_ACCS0), accs1 = OP(IO_ACCS1), accsn = OP(IO_ACCSN);
    cs0 = make_float4(OP(IO_CS0), OP(IO_CS0 + 1), OP(IO_CS0 + 2), OP(IO_CS0 + 3));
    in0 = make_float4(OP(IO_IN0), OP(IO_IN0 + 1), OP(IO_IN0 + 2), OP(IO_IN0 + 3));
    cs1 = make_float4(OP(IO_CS1), OP(IO_CS1 + 1), OP(IO_CS1 + 2), OP(IO_CS1 + 3));
    in1 = make_float4(OP(IO_IN1), OP(IO_IN1 + 1), OP(IO_IN1 + 2), OP(IO_IN1 + 3));
    csn = make_float4(OP(IO_CSN), OP(IO_CSN + 1), OP(IO_CSN + 2), OP(IO_CSN + 3));
    qm0 = OP(IO_QM0), qm1 = OP(IO_QM1), qmn = OP(IO_QMN);
    zr_e_prev = OP(IO_ZRE_P), zr_m_prev = OP(IO_ZRM_P), u_next0 = OP(IO_U_N0);
    zr_e_prevn = OP(IO_ZRE_PN), zr_m_prevn = OP(IO_ZRM_PN), u_nextn = OP(IO_U_NN);
    q1_ii = (has1 && d >= 2u) ? OP(IO_Q1II_A) + OP(IO_Q1II_B) : kNegInf;
    far0 = Acc{OP(IO_FAR0), OP(IO_FAR0 + 1)};
    farn = Acc{OP(IO_FARN), OP(IO_FARN + 1)};
    far1 = Acc{OP(IO_FAR1), OP(IO_FAR1 + 1)};
    mid0 = Acc{OP(IO_MID0), OP(IO_MID0 + 1)};
    mid1 = Acc{OP(IO_MID1), OP(IO_MID1 + 1)};
  } else {
    mbc0 = w0 ? sload(q.m[T_MBC] + row_i + j) : kNegInf;
    mbc1 = (has1 && (w0 || do1)) ? sload(q.m[T_MBC] + row_i + j1) : kNegInf;
    mbcn = (has1 && (w0 || don)) ? sload(q.m[T_MBC] + row_i + ld + j1) : kNegInf;
    hp0 = w0 ? sload(q.m[T_HP] + row_i + j) : kNegInf;
    hp1 = (has1 && do1) ? sload(q.m[T_HP] + row_i + j1) : kNegInf;
    hpn = (has1 && don) ? sload(q.m[T_HP] + row_i + ld + j1) : kNegInf;
    accs0 = w0 ? sload(q.m[T_ACCS] + row_i + j) : 0.f;
    accs1 = (has1 && w0) ? sload(q.m[T_ACCS] + row_i + j1) : 0.f;
    accsn = (has1 && w0) ? sload(q.m[T_ACCS] + row_i + ld + j1) : 0.f;
    cs0 = w0 ? sload4(cs4m + row_i + j) : zero4;
    in0 = w0 ? sload4(in4m + row_i + j) : zero4;
    cs1 = (has1 && do1) ? sload4(cs4m + row_i + j1) : zero4;
    in1 = (has1 && w0) ? sload4(in4m + row_i + j1) : zero4;
    csn = (has1 && don) ? sload4(cs4m + row_i + ld + j1) : zero4;
    qm0 = (w0 && d >= 2) ? sload(q.m[T_QM] + row_i + ld + (j - 1)) : kNegInf;      // Qm(i+1, j-1)
    qm1 = (has1 && do1 && d >= 1) ? sload(q.m[T_QM] + row_i + ld + j) : kNegInf;   // Qm(i+1, j)
    qmn = (has1 && don && d >= 2) ? sload(q.m[T_QM] + row_i + 2 * static_cast<size_t>(ld) + j) : kNegInf;  // Qm(i+2, j)
    if (w0) {
      if (j >= 1) {
        zr_e_prev = sload(q.m[T_ZRE] + col_j - ld + i);
        if (CONTRA) zr_m_prev = sload(q.m[T_ZRM] + col_j - ld + i);
      }
      u_next0 = sload(q.m[T_U] + col_j + i + 1);  // (i+1 == n: the column's pad, -inf)
      if (has1) {
        zr_e_prevn = sload(q.m[T_ZRE] + col_j + i + 1);  // Zr_ext(i+1, j)
        if (CONTRA) zr_m_prevn = sload(q.m[T_ZRM] + col_j + i + 1);
        u_nextn = sload(q.m[T_U] + col_j + ld + i + 2);
      }
      if (!ext_side) {
        zs_a = zs0 ? sload(q.zs + i + 1) : (zs1 ? sload(q.zs + i + 2) : 0.f);  // Z(i+1,n-1) | Z(i+2,n-1)
        zs_last = zs1 ? sload(q.zs + j + 1) : 0.f;                              // Z(n-1,n-1)
        zp1 = (i == 0 && has1) ? sload(q.zp + 1) : 0.f;                         // Z(0,0)
      }
      // the k = i+1 term of sums_multibranch(i,j) (left out of the two-cell product below)
      if (has1 && d >= 2) q1_ii = sload(q.m[T_Q1R] + row_i + i) + sload(q.m[T_ZRM] + col_j + i + 1);
    }
  }
  const bool act0 = mbc0 > kNegInf, act1 = mbc1 > kNegInf, actn = mbcn > kNegInf;
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 8) {  // timing: the uniform operands alone
    const float sum = mbc0 + mbc1 + mbcn + hp0 + hp1 + hpn + accs0 + accs1 + accsn + cs0.x + in0.x + cs1.y +
                      in1.z + csn.w + qm0 + qm1 + qmn + zr_e_prev + zr_m_prev + zr_e_prevn + zr_m_prevn +
                      u_next0 + u_nextn + zs_a + zs_last + zp1 + q1_ii;
    if (t == 0u && sum == 12345.f) q.zp[0] = 1.f;
    return;
  }
#endif

  // [0] [1] [2] closing-pair blocks of (i,j), (i+1,j+1), (i,j+1): one wave each
  float nx = kNegInf;  // (uf) lane 3c + s: near slot s of cell c
  if constexpr (uf) {
    // Near part: the explicit small loops with a + b <= 1 (inner pairs of the last three diagonals) —
    // slot s of cell c in lane 3c + s, all nine at once, gathered with the operands above; the far
    // parts were summed by the previous launch's ahead blocks
    const uint32_t c = lane / 3u;
    const bool mine = ngeo && (c == 0u ? act0 : (c == 1u ? actn : act1));
    nx = mine ? nqb + nsc : kNegInf;
  } else {
  if (act0 && do0)
    pair_block<CONTRA, 64>(b, q, acc[0], i, j, lane, hp0, qm0 + mbc0, cs0,
                           load_win64(q.pk, static_cast<int>(i)), load_win64(q.pk, static_cast<int>(j) - 31));
  if (actn && don)
    pair_block<CONTRA, 64>(b, q, acc[1], i + 1, j1, lane, hpn, qmn + mbcn, csn,
                           load_win64(q.pk, static_cast<int>(i) + 1), load_win64(q.pk, static_cast<int>(j) - 30));
  if (act1 && do1)
    pair_block<CONTRA, 64>(b, q, acc[2], i, j1, lane, hp1, qm1 + mbc1, cs1,
                           load_win64(q.pk, static_cast<int>(i)), load_win64(q.pk, static_cast<int>(j) - 30));
  }
  if constexpr (!uf) products();
  if (t == 0u) acc_add(acc[3], q1_ii);
  if (thr != 0u && w0) {
    const Acc m0 = uf ? mid0 : load_mid(q, b.ring, 0u, d, i);
    const Acc m1 = uf ? mid1 : (has1 ? load_mid(q, b.ring, 0u, d + 1u, i) : acc_empty());
    if (t == 0u) {
      acc_merge(acc[3], m0);
      acc_merge(acc[4], m1);
    }
  }
  // (the sums_external accumulators [5..8] live only in row 0 and in the column n-1 groups)
  if (!cell_reduce<NA, TPC>(acc, red, ((row0 || zs0 || zs1) ? 0x1ffu : 0x1fu) & (uf ? ~7u : ~0u))) return;
  if constexpr (uf) {
    // the blocks' sums: far part, hairpin, multibranch term, three near slots (uniform values)
    auto lanev = [&](uint32_t l) {
      return __uint_as_float(static_cast<uint32_t>(
          __builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(nx)), static_cast<int>(l))));
    };
    acc[0] = far0;
    acc_add4(acc[0], hp0, qm0 + mbc0, lanev(0), lanev(1));
    acc_add(acc[0], lanev(2));
    acc[1] = farn;
    acc_add4(acc[1], hpn, qmn + mbcn, lanev(3), lanev(4));
    acc_add(acc[1], lanev(5));
    acc[2] = far1;
    acc_add4(acc[2], hp1, qm1 + mbc1, lanev(6), lanev(7));
    acc_add(acc[2], lanev(8));
  }
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 16) {  // timing: no epilogue
    if (t == 0u && acc[0].s + acc[1].s + acc[2].s + acc[3].s + acc[4].s == 12345.f) q.zp[0] = 1.f;
    return;
  }
#endif

  const float ext_bp = CONTRA ? b.params->contra.external_score_basepair : 0.f;
  const float ext_un = CONTRA ? b.params->contra.external_score_unpair : 0.f;
  const float mb_bp = CONTRA ? b.params->contra.multibranch_score_basepair
                             : b.params->turner.coeff_num_branches;
  const float mb_un = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  const bool st = t == 0u;  // the lane that stores
  float* __restrict__ qx = q.m[T_X4];
  const size_t xsz = static_cast<size_t>(q.m[1] - q.m[0]);
  auto store_x4 = [&](size_t o, float v, const float4& w) {
    qx[o] = v + w.x;
    qx[xsz + o] = v + w.y;
    qx[2u * xsz + o] = v + w.z;
    qx[3u * xsz + o] = v + w.w;
  };
  // ---- cell (i,j)
  float qa0 = kNegInf;
  if (act0) {
    const float qb = acc_value(acc[0]);
    if (qb > kNegInf) {
      qa0 = qb + accs0;
      if (st) {
        q.m[T_QB][row_i + j] = qb;
        q.m[T_QA][row_i + j] = qa0;
        store_x4(row_i + j, qb, in0);
      }
    }
  }
  const float zr_e0 = lse2(zr_e_prev + ext_un, qa0 + ext_bp);
  const float zr_m0 = CONTRA ? lse2(zr_m_prev + mb_un, qa0 + mb_bp) : zr_e0 + mb_bp;
  const float u0 = lse2(u_next0 + mb_un, zr_m0);
  const float qmv0 = acc_value(acc[3]);
  const float q1_0 = lse2(u0, qmv0);
  if (st) {
    q.m[T_ZRE][col_j + i] = zr_e0;
    q.m[T_ZRM][col_j + i] = zr_m0;
    q.m[T_U][col_j + i] = u0;
    q.m[T_QM][row_i + j] = qmv0;
    q.m[T_Q1R][row_i + j] = q1_0;
    q.m[T_Q1C][col_j + i] = q1_0;
  }
  if (row0) {
    // sums_external[0][j] (352-363 / 487-498)
    Acc z = acc[5];
    acc_add(z, zr_e0);  // k = 0: Z(0,-1) = 0
    acc_add(z, CONTRA ? ext_un * static_cast<float>(j + 1) : 0.f);
    if (st) q.zp[j + 1] = acc_value(z);
  }
  if (zs0) {
    Acc z = acc[7];
    z.m += ext_bp;             // every product term carries the pair's external_score_basepair
    acc_add(z, qa0 + ext_bp);  // l = n-1: Z(n,n-1) = 0
    acc_add(z, zs_a + ext_un);
    if (st) q.zs[i] = acc_value(z);
  }
  if (!has1) return;
  // ---- neighbour (i+1,j+1): closing pair -> Zr -> U, not stored (its own group does)
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
        store_x4(row_i + j1, qb, in1);
      }
    }
  }
  const float zr_e1 = lse2(zr_e0 + ext_un, qa1 + ext_bp);
  const float zr_m1 = CONTRA ? lse2(zr_m0 + mb_un, qa1 + mb_bp) : zr_e1 + mb_bp;
  const float u1 = lse2(un + mb_un, zr_m1);
  const float qmv1 = acc_value(acc[4]);
  const float q1_1 = lse2(u1, qmv1);
  if (st) {
    q.m[T_ZRE][col_j1 + i] = zr_e1;
    q.m[T_ZRM][col_j1 + i] = zr_m1;
    q.m[T_U][col_j1 + i] = u1;
    q.m[T_QM][row_i + j1] = qmv1;
    q.m[T_Q1R][row_i + j1] = q1_1;
    q.m[T_Q1C][col_j1 + i] = q1_1;
  }
  if (row0) {
    Acc z = acc[6];
    acc_add(z, zr_e1);        // k = 0
    acc_add(z, zr_en + zp1);  // k = 1: Zr_ext(1,j+1) + Z(0,0)
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
    acc_add(z, (qa0 + ext_bp) + zs_last);  // l = j = n-2: Z(n-1,n-1)
    acc_add(z, qa1 + ext_bp);              // l = n-1
    acc_add(z, zs_n + ext_un);
    if (st) q.zs[i] = acc_value(z);
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
                                              const float* __restrict__ Qb, uint32_t len0,
                                              uint32_t lenn, bool do_n, bool do_1, uint32_t t) {
  // (lenn <= len0: where the neighbour's stream ends)
  for (uint32_t k = t; k < len0; k += 4u * TPC) {
    float x0[4], xn[4], x1[4], wi[4], qa[4], wm[4], qb[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t ku = k + static_cast<uint32_t>(u) * TPC;
      const bool v = ku < len0, vn = ku < lenn && do_n, vb = v && do_1 && ku >= 1u;
      const uint32_t kv = v ? ku : 0u;
      wi[u] = Wi[kv], qa[u] = Qa[kv];
      wm[u] = do_n ? Wm[vn ? ku : 0u] : kNegInf;  // (do_n, do_1: uniform)
      qb[u] = do_1 ? Qb[vb ? ku : 1u] : kNegInf;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t ku = k + static_cast<uint32_t>(u) * TPC;
      const bool v = ku < len0, vn = ku < lenn && do_n, vb = v && do_1 && ku >= 1u;
      x0[u] = v ? wi[u] + qa[u] : kNegInf;
      xn[u] = vn ? wm[u] + qa[u] : kNegInf;
      x1[u] = vb ? wi[u] + qb[u] : kNegInf;
    }
    acc_add4(pm0, x0[0], x0[1], x0[2], x0[3]);
    acc_add4(pmn, xn[0], xn[1], xn[2], xn[3]);
    acc_add4(pm1, x1[0], x1[1], x1[2], x1[3]);
  }
}

// diagonals d+1 (cell (i,j+1)) and d (cell (i,j)) of the outside sweep
// `thr` (outside): banded cells.  The terms of probs_multibranch and of L_e (cases one / three)
// whose OUTSIDE operand (W(i,k), R(k,j)) spans at least thr were summed by k_tree_mid before
// the band above this one started; the launch adds the nearer ones and merges.
template <bool CONTRA, int TPC, bool UF>
__global__ void __launch_bounds__(TPC < 256 ? 256 : TPC) k_tree_outside2(float* hbase, uint64_t hmsz, uint32_t hn, uint32_t hld,
                                                                         uint32_t d, uint32_t thr, uint32_t use_one,
                                                                         TreeBatch b, int single, Ahead ah) {
  static_assert(!UF || TPC == 64, "far parts from the ring: one wave per cell pair");
  constexpr int BLOCK = TPC < 256 ? 256 : TPC;
  constexpr int NA = 7;
  __shared__ float red[BLOCK / 64][NA][2];
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 4) return;
#endif
  const TSeq q = load_tseq_hot(b, blockIdx.y, hbase, hmsz, hn, hld, use_one);
  const uint32_t n = q.n, ld = q.ld;
  const uint32_t bx = role_block(ah);
  if (bx >= ah.main_blocks) {
    // ---- the NEXT launch's enclosing 2-loops, far part (see Ahead): one wave per row i takes
    // the cells (i, i+nd0) and (i, i+nd0+1)
#ifdef RNAMC_DEBUG_KNOBS
    if (b.debug & 1) return;
#endif
    __builtin_amdgcn_s_setprio(1);
    const bool split = (ah.flags & 2u) != 0u;
    const uint32_t wid = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
        static_cast<int>((bx - ah.main_blocks) * (BLOCK / 64) + threadIdx.x / 64)));
    const uint32_t i = split ? wid >> 1 : wid;
    const bool doA = !split || (wid & 1u) == 0u, doB = !split || (wid & 1u) != 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const size_t row_i = static_cast<size_t>(i) * ld;
    const uint32_t j0 = i + ah.nd0;
    if (j0 >= n) return;
    const uint64_t wi = load_win64(q.pk, static_cast<int>(i) - 31);
    const bool two = doB && ah.nd_count > 1u && j0 + 1u < n;
    const float qbA = doA ? sload(q.m[T_QB] + row_i + j0) : kNegInf;
    const float qbB = two ? sload(q.m[T_QB] + row_i + j0 + 1u) : kNegInf;
    const float4 inA = sload4(reinterpret_cast<const float4*>(q.m[T_IN4]) + row_i + j0);
    const float4 inB = sload4(reinterpret_cast<const float4*>(q.m[T_IN4]) + row_i + j0 + (two ? 1u : 0u));
    const uint64_t wjA = load_win64(q.pk, static_cast<int>(j0));
    const uint64_t wjB = load_win64(q.pk, static_cast<int>(j0) + 1);
    Acc fA = acc_empty(), fB = acc_empty();
    if (qbA > kNegInf) outer_far<CONTRA>(b, q, fA, i, j0, lane, qbA, inA, wi, wjA);
    if (qbB > kNegInf) outer_far<CONTRA>(b, q, fB, i, j0 + 1u, lane, qbB, inB, wi, wjB);
    if (qbA > kNegInf) fA = wave_reduce(fA);
    if (qbB > kNegInf) fB = wave_reduce(fB);
    if (lane == 0u) {
      if (doA) q.far[static_cast<size_t>(ah.nd0 & 3u) * q.vec + i] = make_float2(fA.m, fA.s);
      if (two) q.far[static_cast<size_t>((ah.nd0 + 1u) & 3u) * q.vec + i] = make_float2(fB.m, fB.s);
    }
    return;
  }
  const uint32_t i = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
      static_cast<int>(bx * (BLOCK / TPC) + threadIdx.x / TPC)));
  if (i + d >= n) return;
  __builtin_amdgcn_s_setprio(3);  // (as in the inside kernel)
  constexpr bool uf = UF;
  const uint32_t j = i + d, j1 = j + 1u;
  const bool has1 = !single && j1 < n;
  const uint32_t t = threadIdx.x % TPC;
  const float* __restrict__ qb_r = q.m[T_QB];
  const size_t row_i = static_cast<size_t>(i) * ld, col_j = static_cast<size_t>(j) * ld;
  const float* __restrict__ w_r = q.m[T_ZRE];   // W = (P + mbclose) - Qb, row-major
  float* __restrict__ r_c = q.m[T_ZRM];         // R = Pm (+) Pm2, column-major
  float* __restrict__ pm2_r = q.m[T_QM];        // probs_multibranch2, row-major
  float* __restrict__ sp_c = q.m[T_U];          // sp_c(i,j) = (+)_{k<=i} Pm(k,j) [+ unpaired], column-major
  const float4* __restrict__ cs4m = reinterpret_cast<const float4*>(q.m[T_CS4]);
  const float4* __restrict__ in4m = reinterpret_cast<const float4*>(q.m[T_IN4]);

  // roles as in the inside kernel: the group's first two waves take one block of enclosing
  // 2-loops each, the first wave alone the scalar operands and the epilogue
  const uint32_t wv = TPC == 64 ? 0u : static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(t >> 6)));
  const bool w0 = wv == 0u;
  const bool do0 = w0, do1 = uf ? w0 : (TPC == 64 || wv == 1u);
  const uint32_t lane = t & 63u;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t jt = has1 ? j1 : j;  // the upper cell of this group
  float qb0 = kNegInf, qb1 = kNegInf;
  float pm2_nextt = kNegInf, w_nextt = kNegInf, sp_prev1 = kNegInf, sp_prev2 = kNegInf;
  float ztot = 0.f, zpi = 0.f, qa0 = kNegInf, mbc0 = kNegInf, zsj0 = 0.f, qa1 = kNegInf, mbc1 = kNegInf, zsj1 = 0.f;
  float4 cs0 = zero4, cs1 = zero4, in0 = zero4, in1 = zero4;
  Acc far0 = acc_empty(), far1 = acc_empty();
  Acc mp0 = acc_empty(), mpn = acc_empty(), mp1 = acc_empty(), me0 = acc_empty(), me1 = acc_empty();
  float nqb = kNegInf, npk = kNegInf, nsc = 0.f;  // (uf) lane 3c + s: closing pair of near slot s of cell c
  bool ngeo = false;
  if constexpr (uf) {
    // (one vector gather for the ~46 uniform operands: gather_operand above)
    const OpCtx oc{hbase_of(q), static_cast<uint64_t>(q.m[1] - q.m[0]), q.far, q.mid, q.zp, q.zs, i, j, d, n, ld, q.vec, b.ring,
                   has1, CONTRA, thr != 0u};
    const float ov = gather_operand(kOutOps[lane], oc);
    {
      // the near enclosing pairs (closing pairs of the last three diagonals): slot s of cell c in lane
      // 3c + s; whether the cell is a pair at all is applied to the value below
      const uint32_t c = lane / 3u, sl = lane - 3u * c;
      const uint32_t cj = c == 0u ? j : j1;
      uint32_t sa, sb;
      Special<CONTRA>::slot(sl < kNear ? sl : 0u, sa, sb);
      ngeo = lane < 6u && (c == 0u || has1) && sa < i && cj + 1u + sb < n;
      if (ngeo) {
        const uint32_t k = i - 1u - sa, l = cj + 1u + sb;
        const size_t o = static_cast<size_t>(k) * ld + l;
        nqb = q.m[T_QB][o];
        npk = q.out[tri_off(n, l - k) + k];
        nsc = q.m[T_NEAR4][4u * o + sl];  // (the closing pair's static: slot sl of (k,l))
      }
    }
    auto OP = [&](int x) { return lane_value(ov, x); };
    qb0 = OP(OO_QB0), qb1 = OP(OO_QB1);
    pm2_nextt = has1 ? OP(OO_PM2_A) : OP(OO_PM2_B);
    w_nextt = has1 ? OP(OO_W_A) : OP(OO_W_B);
    sp_prev1 = OP(OO_SP1), sp_prev2 = OP(OO_SP2);
    ztot = OP(OO_ZTOT), zpi = OP(OO_ZPI), zsj0 = OP(OO_ZSJ0), zsj1 = OP(OO_ZSJ1);
    qa0 = OP(OO_QA0), mbc0 = OP(OO_MBC0), qa1 = OP(OO_QA1), mbc1 = OP(OO_MBC1);
    cs0 = make_float4(OP(OO_CS0), OP(OO_CS0 + 1), OP(OO_CS0 + 2), OP(OO_CS0 + 3));
    cs1 = make_float4(OP(OO_CS1), OP(OO_CS1 + 1), OP(OO_CS1 + 2), OP(OO_CS1 + 3));
    in0 = make_float4(OP(OO_IN0), OP(OO_IN0 + 1), OP(OO_IN0 + 2), OP(OO_IN0 + 3));
    in1 = make_float4(OP(OO_IN1), OP(OO_IN1 + 1), OP(OO_IN1 + 2), OP(OO_IN1 + 3));
    far0 = Acc{OP(OO_FAR0), OP(OO_FAR0 + 1)};
    far1 = Acc{OP(OO_FAR1), OP(OO_FAR1 + 1)};
    mp0 = Acc{OP(OO_P0), OP(OO_P0 + 1)};
    mpn = Acc{OP(OO_PN), OP(OO_PN + 1)};
    mp1 = Acc{OP(OO_P1), OP(OO_P1 + 1)};
    me0 = Acc{OP(OO_E0), OP(OO_E0 + 1)};
    me1 = Acc{OP(OO_E1), OP(OO_E1 + 1)};
  } else {
    qb0 = sload(qb_r + row_i + j);
    qb1 = has1 ? sload(qb_r + row_i + j1) : kNegInf;
    if (w0) {  // (all from diagonals >= d+2, or statics)
      if (jt + 1 < n) {
        pm2_nextt = sload(pm2_r + row_i + jt + 1);  // Pm2(i, jt+1)
        w_nextt = sload(w_r + row_i + jt + 1);      // W(i, jt+1)
      }
      if (i >= 1 && has1) sp_prev1 = sload(sp_c + col_j + ld + i - 1);  // prefix of column j+1 up to row i-1
      if (i >= 2) sp_prev2 = sload(sp_c + col_j + i - 2);               // prefix of column j up to row i-2
      ztot = sload(q.zp + n);
      zpi = sload(q.zp + i);
      qa0 = sload(q.m[T_QA] + row_i + j);
      mbc0 = sload(q.m[T_MBC] + row_i + j);
      zsj0 = sload(q.zs + j + 1);
      cs0 = sload4(cs4m + row_i + j);
      if (has1) {
        qa1 = sload(q.m[T_QA] + row_i + j1);
        mbc1 = sload(q.m[T_MBC] + row_i + j1);
        zsj1 = sload(q.zs + j1 + 1);
        cs1 = sload4(cs4m + row_i + j1);
      }
    }
    in0 = do0 ? sload4(in4m + row_i + j) : zero4;
    in1 = (has1 && do1) ? sload4(in4m + row_i + j1) : zero4;
  }
  const bool paired0 = qb0 > kNegInf, paired1 = qb1 > kNegInf;  // (uniform)

  Acc acc[NA];
#pragma unroll
  for (int x = 0; x < NA; x++) acc[x] = acc_empty();
  // [3] [4] enclosing 2-loops of (i,j) and (i,j+1): one wave each
  float nx = kNegInf;  // (uf) lane 3c + s: near slot s of cell c
  if constexpr (uf) {
    // near part (the enclosing pairs of the last three diagonals), gathered with the operands above
    const uint32_t c = lane / 3u;
    const float qbc = c == 0u ? qb0 : qb1;
    if (ngeo && (c == 0u ? paired0 : paired1) && nqb > kNegInf) nx = ((npk + qbc) - nqb) + nsc;
  } else {
  if (paired0 && do0)
    outer_block<CONTRA, 64>(b, q, acc[3], i, j, lane, qb0, in0, load_win64(q.pk, static_cast<int>(i) - 31),
                            load_win64(q.pk, static_cast<int>(j)));
  if (paired1 && do1)
    outer_block<CONTRA, 64>(b, q, acc[4], i, j1, lane, qb1, in1, load_win64(q.pk, static_cast<int>(i) - 31),
                            load_win64(q.pk, static_cast<int>(j) + 1));
  }
  // [0] Pm(i,j), [1] Pm(i-1,j), [2] Pm(i,j+1): k = j+1 .. n-1 (the k = j+1 term of [0] reads
  // W(i,j+1) of this launch next to Q1(j+1,j) = -inf: harmless)
#ifdef RNAMC_DEBUG_KNOBS
  if (!(b.debug & 2))
#endif
  {
    if (n - 1 - j >= 2) {
      // idx: W(i | i-1, j+1+idx) [span d+1+idx | d+2+idx]: the launch's part ends below span thr
      const uint32_t len0 = n - 1 - j;
      const uint32_t lim0 = thr != 0u ? min(len0, thr - 1u - d) : len0;
      const uint32_t limn = thr != 0u ? min(len0, thr - 2u - d) : len0;
      acc_product_3<TPC>(acc[0], acc[1], acc[2], w_r + row_i + j1, w_r + row_i - ld + j1,
                         q.m[T_Q1R] + static_cast<size_t>(j1) * ld + j,
                         q.m[T_Q1R] + static_cast<size_t>(j1 + 1) * ld + j, lim0, limn, i >= 1, has1, t);
    }
    // [5] [6] L_e cases one and three of (i,j) and (i,j+1): k = 0 .. i-1, Q1(k+1,i-1) + R(k,.)
    // (banded: R(k, j | j+1) of span below thr, k >= j - thr + 1 | j - thr + 2)
    if (i >= 1 && (paired0 || paired1)) {
      const uint32_t base = (thr != 0u && j + 1u > thr) ? j + 1u - thr : 0u;
      const uint32_t lo1 = (thr != 0u && j + 1u >= thr) ? 1u : 0u;
      const float* __restrict__ A = q.m[T_Q1C] + static_cast<size_t>(i - 1) * ld + 1 + base;
      if (has1)
        acc_product_2b_lo<TPC>(acc[5], acc[6], A, r_c + col_j + base, r_c + col_j + ld + base,
                               i - 1 - base, i - base, lo1, t);
      else
        acc_product<TPC>(acc[5], A, r_c + col_j + base, i - 1 - base, t);
    }
  }
  if (thr != 0u && w0) {
    // (the ring rows of diagonals d, d+1 belong to this band; cells past the matrix are never read)
    const Acc p0 = uf ? mp0 : load_mid(q, b.ring, 1u, d, i);
    const Acc pn = uf ? mpn : ((i >= 1 && j < n - 1) ? load_mid(q, b.ring, 1u, d + 1u, i - 1u) : acc_empty());
    const Acc p1 = uf ? mp1 : (has1 ? load_mid(q, b.ring, 1u, d + 1u, i) : acc_empty());
    const Acc e0 = uf ? me0 : load_mid(q, b.ring, 2u, d, i);
    const Acc e1 = uf ? me1 : (has1 ? load_mid(q, b.ring, 2u, d + 1u, i) : acc_empty());
    if (t == 0u) {
      acc_merge(acc[0], p0);
      acc_merge(acc[1], pn);
      acc_merge(acc[2], p1);
      acc_merge(acc[5], e0);
      acc_merge(acc[6], e1);
    }
  }
  if (!cell_reduce<NA, TPC>(acc, red, uf ? 0x67u : 0x7fu)) return;
  if constexpr (uf) {
    auto lanev = [&](uint32_t l) {
      return __uint_as_float(static_cast<uint32_t>(
          __builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(nx)), static_cast<int>(l))));
    };
    acc[3] = far0;
    acc_add4(acc[3], lanev(0), lanev(1), lanev(2), kNegInf);
    acc[4] = far1;
    acc_add4(acc[4], lanev(3), lanev(4), lanev(5), kNegInf);
  }

  const bool st = t == 0u;
  const float mb_un = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  const float ext_bp = CONTRA ? b.params->contra.external_score_basepair : 0.f;
  const float abr = CONTRA ? b.params->contra.multibranch_score_basepair
                           : b.params->turner.coeff_num_branches;
  float* __restrict__ px = q.m[T_X4];
  const size_t xsz = static_cast<size_t>(q.m[1] - q.m[0]);
  auto store_x4 = [&](size_t o, float v, const float4& w) {
    px[o] = v + w.x;
    px[xsz + o] = v + w.y;
    px[2u * xsz + o] = v + w.z;
    px[3u * xsz + o] = v + w.w;
  };
  // ---- cell (i,j+1)
  float w1 = kNegInf, pm2_1 = kNegInf;
  if (has1) {
    const float pm1 = acc_value(acc[2]);
    pm2_1 = lse2(pm2_nextt + mb_un, w_nextt);
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
          const float pq = lp - qb1;
          store_x4(row_i + j1, pq, cs1);
        }
      }
    }
  } else {
    // the single cell: Pm2 and W of its right neighbour come from memory
    pm2_1 = pm2_nextt;
    w1 = w_nextt;
  }
  // ---- cell (i,j); the prefix of column j up to row i-1 needs Pm(i-1,j) (a cell of this
  // launch when has1, recomputed here either way)
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
    const float pq = lp - qb0;
    store_x4(row_i + j, pq, cs0);
  }
}

// ----------------------------------------------------------------------------
// Banded mid-field of the cubic products.  A launch of the sweep is a chain of dependent round
// trips, and a cell's product walks d terms of it; but of the d terms of sums_multibranch(i,j)
// only those with an operand from the last two BANDS of diagonals (band = kBand diagonals) depend
// on recent launches.  All others — both operands of span below thr = (band - 1) * width (inside),
// resp. the outside operand of span at least thr = (band + 2) * width (outside) — are final one
// whole band before the cell's own launch.  k_tree_mid sums them for every cell of a band as
// a tiled (logsumexp,+) matrix product on a second stream while the previous band sweeps: the
// operands of an 8 x 32 tile of cells (rows i, diagonals d) are staged through LDS in pieces of 16
// k and every lane folds four cells (one row, four diagonals) off them (a term costs ~10 VALU
// slots and 5 B of LDS reads, no HBM round trip).  The sweep's launches then walk at most 4 band
// widths of terms per product and merge the cell's {max, sum} pair from the ring.
//   prod 0 (inside)  C(i,j) = (+)_{k = j-thr+1 .. i+thr}  Q1(i,k-1) + Zr_mb(k,j)
//   prod 1 (outside) C(i,j) = (+)_{k = i+thr .. n-1}      W(i,k)    + Q1(j+1,k-1)
//   prod 2 (outside) C(i,j) = (+)_{k = 0 .. j-thr}        Q1(k+1,i-1) + R(k,j)
// A workgroup is eight waves = eight interleaved sets of k pieces of ONE tile, merged through LDS
// at the end (fixed order: results are deterministic).
constexpr int kMidTI = 8, kMidTD = 32, kMidKC = 16, kMidRow = 20, kMidWaves = 8;
constexpr int kMidRows = kMidTI + kMidTI + kMidTD - 1;  // 8 A rows + 39 B columns
constexpr int kMidPf = (kMidRows * kMidKC + 63) / 64;   // staged elements per lane and piece

// (at most 84 VGPRs: six waves per SIMD, so that the sweep's launches, ~85 VGPRs, keep four of
// their five waves per SIMD while a mid-field kernel is resident)
__global__ void __launch_bounds__(64 * kMidWaves) __attribute__((amdgpu_waves_per_eu(6, 6)))
k_tree_mid(TreeBatch b, uint32_t dlo, uint32_t dhi, uint32_t thr, int outside, uint32_t tiles_i,
           uint32_t tiles_z) {
  __shared__ float stage[kMidWaves][(kMidRows + 1) * kMidRow];
  const TSeq q = load_tseq(b, blockIdx.y);
  const int n = static_cast<int>(q.n), ld = static_cast<int>(q.ld);
  // a fixed number of workgroups walks the band's tiles (the grid holds about two waves per
  // SIMD, whatever the band's size: the sweep's launches keep the rest of the chip)
  for (uint32_t tile = blockIdx.x; tile < tiles_i * tiles_z; tile += gridDim.x) {
  const uint32_t tz = tile / tiles_i, tx = tile % tiles_i;
  const int prod = outside ? 1 + static_cast<int>(tz & 1u) : 0;
  const int d0 = static_cast<int>(dlo) + kMidTD * static_cast<int>(outside ? tz >> 1 : tz);
  const int i0 = kMidTI * static_cast<int>(tx);
  const int dtop = min(static_cast<int>(dhi), n - 1);
  if (d0 > dtop || i0 + d0 >= n) continue;  // (uniform: no cell of this tile exists)
  const int T = static_cast<int>(thr);
  const int wave = static_cast<int>(threadIdx.x >> 6), lane = static_cast<int>(threadIdx.x & 63u);
  const int li = lane & 7, lg = lane >> 3;

  // operand rows: a uniform 64-bit base per matrix (scalar registers) + 32-bit float offsets INSIDE
  // the matrix (ld * n < 2^32 for every n the index type admits), so that a sequence's 27
  // matrices may span more than 2^32 floats (n = 16 384: 29 GB)
  const float* __restrict__ baseA;
  const float* __restrict__ baseB;
  uint32_t offA, offB;  // row i0 of A / column (i0 + d0) of B, at k = 0 (wraps for the "- 1" forms: added back below)
  int kmin;             // smallest k whose operands exist
  if (prod == 0) {
    baseA = q.m[T_Q1R];
    baseB = q.m[T_ZRM];
    offA = static_cast<uint32_t>(i0) * ld - 1u;
    offB = static_cast<uint32_t>(i0 + d0) * ld;
    kmin = 1;
  } else if (prod == 1) {
    baseA = q.m[T_ZRE];
    baseB = q.m[T_Q1R];
    offA = static_cast<uint32_t>(i0) * ld;
    offB = static_cast<uint32_t>(i0 + d0 + 1) * ld - 1u;
    kmin = 1;
  } else {
    baseA = q.m[T_Q1C];
    baseB = q.m[T_ZRM];
    offA = static_cast<uint32_t>(i0 - 1) * ld + 1u;  // (row i0 - 1; i0 = 0: masked below)
    offB = static_cast<uint32_t>(i0 + d0) * ld;
    kmin = 0;
  }
  // rows of the stage that exist: A row r <-> i = i0 + r, B row r <-> j = i0 + d0 + r
  uint64_t rowmask = 0;
  for (int r = 0; r < kMidTI; r++) {
    const int i = i0 + r;
    if (i < n && (prod != 2 || i >= 1)) rowmask |= 1ull << r;
  }
  for (int r = 0; r < kMidTI + kMidTD - 1; r++) {
    const int j = i0 + d0 + r;
    if (prod == 1 ? j + 1 < n : j < n) rowmask |= 1ull << (kMidTI + r);
  }
  // the tile's k range, and the part of it every cell of the tile takes whole
  int klo, khi, ilo, ihi;
  if (prod == 0) {
    klo = i0 + d0 - T + 1;
    khi = i0 + kMidTI - 1 + T;
    ilo = (i0 + kMidTI - 1) + (d0 + kMidTD - 1) - T + 1;
    ihi = i0 + T;
  } else if (prod == 1) {
    klo = i0 + T;
    khi = n - 1;
    ilo = i0 + kMidTI - 1 + T;
    ihi = n - 1;
  } else {
    klo = 0;
    khi = (i0 + kMidTI - 1) + (d0 + kMidTD - 1) - T;
    ilo = 0;
    ihi = i0 + d0 - T;
  }
  klo = max(klo, kmin);
  khi = min(khi, n - 1);

  // the lane's cells: row i0 + li, diagonals d0 + lg + 8 c (c < 4); B column li + lg + 8 c
  const int ci = i0 + li;
  Acc acc[4];
#pragma unroll
  for (int c = 0; c < 4; c++) acc[c] = acc_empty();

  float* __restrict__ st = stage[wave];
  const int nch = klo <= khi ? (khi - klo) / kMidKC + 1 : 0;
  // piece `ch` of the operands into registers: element e = lane + 64 u <-> row e / 16, k e % 16;
  // rows 4u .. 4u+3 are A rows for u < kMidTI / 4, B columns beyond: lane part + uniform part
  const uint32_t lrow = static_cast<uint32_t>(lane) >> 4, lk = static_cast<uint32_t>(lane) & 15u;
  const uint32_t vA = offA + lrow * static_cast<uint32_t>(ld) + lk;
  const uint32_t vB = offB + lrow * static_cast<uint32_t>(ld) + lk;
  uint32_t lmask = 0;  // bit u: the lane's row of step u exists
#pragma unroll
  for (int u = 0; u < kMidPf; u++) {
    const uint32_t row = lrow + 4u * u;
    if (row < kMidRows && ((rowmask >> row) & 1ull) != 0ull) lmask |= 1u << u;
  }
  float pf[kMidPf];
  auto fetch = [&](int ch) {
    const uint32_t kk = static_cast<uint32_t>(klo + ch * kMidKC);  // (uniform)
#pragma unroll
    for (int u = 0; u < kMidPf; u++) {
      const uint32_t uni = (u < kMidTI / 4 ? 4u * u : 4u * (u - kMidTI / 4)) * static_cast<uint32_t>(ld) + kk;
      const uint32_t off = (u < kMidTI / 4 ? vA : vB) + uni;  // (mod 2^32: k >= kmin undoes the "- 1")
      pf[u] = ((lmask >> u) & 1u) ? (u < kMidTI / 4 ? baseA : baseB)[off] : kNegInf;
    }
  };
  if (wave < nch) fetch(wave);
  for (int ch = wave; ch < nch; ch += kMidWaves) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < kMidPf; u++) {
      const int row = (lane >> 4) + 4 * u;  // (the stage has room for row kMidRows too)
      st[row * kMidRow + (lane & 15)] = pf[u];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (ch + kMidWaves < nch) fetch(ch + kMidWaves);
    const int kc = klo + ch * kMidKC;
    const bool whole = kc >= ilo && kc + kMidKC - 1 <= ihi;  // (uniform)
    if (whole) {
#pragma unroll 1
      for (int k4 = 0; k4 < kMidKC / 4; k4++) {
        const float4 a = *reinterpret_cast<const float4*>(st + li * kMidRow + 4 * k4);
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const float4 bq = *reinterpret_cast<const float4*>(st + (kMidTI + li + lg + 8 * c) * kMidRow + 4 * k4);
          acc_add4(acc[c], a.x + bq.x, a.y + bq.y, a.z + bq.z, a.w + bq.w);
        }
      }
    } else {
#pragma unroll 1
      for (int k4 = 0; k4 < kMidKC / 4; k4++) {
        const float4 a = *reinterpret_cast<const float4*>(st + li * kMidRow + 4 * k4);
        const int k = kc + 4 * k4;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const float4 bq = *reinterpret_cast<const float4*>(st + (kMidTI + li + lg + 8 * c) * kMidRow + 4 * k4);
          // the cell's own k range as (first k, last k - first k); no terms: (2^30, 0)
          const int d = d0 + lg + 8 * c, j = ci + d;
          int lo = prod == 0 ? j - T + 1 : (prod == 1 ? ci + T : 0);
          const int hi = prod == 0 ? ci + T : (prod == 1 ? n - 1 : j - T);
          lo = max(lo, kmin);
          const bool ok = d <= dtop && j < n && lo <= hi;
          const uint32_t rel = static_cast<uint32_t>(k - (ok ? lo : (1 << 30)));
          const uint32_t len = ok ? static_cast<uint32_t>(hi - lo) : 0u;
          const float x0 = rel <= len ? a.x + bq.x : kNegInf;
          const float x1 = rel + 1u <= len ? a.y + bq.y : kNegInf;
          const float x2 = rel + 2u <= len ? a.z + bq.z : kNegInf;
          const float x3 = rel + 3u <= len ? a.w + bq.w : kNegInf;
          acc_add4(acc[c], x0, x1, x2, x3);
        }
      }
    }
  }
  // merge the eight waves' partial sums: wave w, lanes 32 (w & 1) .. + 31 fold register cell
  // w >> 1 of the 64 lanes' cells, two threads per ... (kept simple: waves 0..3 fold one cell each)
  __syncthreads();
  float* __restrict__ red = &stage[0][0];  // [wave][cell][lane][2]: 8 * 4 * 64 * 2 floats = 16 KB
#pragma unroll
  for (int c = 0; c < 4; c++) {
    float* p = red + ((wave * 4 + c) * 64 + lane) * 2;
    p[0] = acc[c].m;
    p[1] = acc[c].s;
  }
  __syncthreads();
  if (wave < 4) {
    Acc tot = acc_empty();
#pragma unroll
    for (int w = 0; w < kMidWaves; w++) {
      const float* p = red + ((w * 4 + wave) * 64 + lane) * 2;
      acc_merge(tot, Acc{p[0], p[1]});
    }
    const int d = d0 + lg + 8 * wave;
    if (d <= dtop && ci + d < n)
      q.mid[(static_cast<size_t>(prod) * b.ring + static_cast<uint32_t>(d) % b.ring) * q.vec + ci] =
          make_float2(tot.m, tot.s);
  }
  __syncthreads();  // (the next tile's pieces overwrite the exchange area)
  }
}

// sums_external of a banded sweep: its first row Z(0,j) (prefix vector zp; 352-363 / 487-498)
// and last column Z(i,n-1) (suffix vector zs, leftmost-pair decomposition) are read by the
// outside sweep only, and step d of either needs just diagonal d of the inside sweep.  Inside a
// launch each is ONE cell with a sum of d terms — the longest chain of the launch once the products
// are banded.  This kernel walks them a band behind the sweep instead, beside it: workgroup 0 the
// row, workgroup 1 the column, one step per diagonal d in [dlo, dhi], 1024 lanes per sum, the vector
// in LDS, the matrix operand of the next step in flight while this one is reduced.
//   zp[j+1] = (+)_{k=0..j} (Zr_ext(k,j) + zp[k])  (+)  unpaired          (j = d; zp[0] = 0)
//   zs[i]   = ((+)_{l=i+1..n-1} Qa(i,l) + zs[l+1]) + ext_bp  (+)  (zs[i+1] + ext_un)   (i = n-1-d)
template <bool CONTRA>
__global__ void __launch_bounds__(1024) k_tree_ext(TreeBatch b, uint32_t dlo, uint32_t dhi, int use_lds) {
  extern __shared__ float lds_vec[];  // zp[0 .. n] | zs[0 .. n] shifted so that the walk reads vec[k]
  __shared__ float red[16][2];
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  if (dlo >= n) return;
  const uint32_t dtop = min(dhi, n - 1u);
  const bool col = blockIdx.x == 1u;
  const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63u;
  const float ext_bp = CONTRA ? b.params->contra.external_score_basepair : 0.f;
  const float ext_un = CONTRA ? b.params->contra.external_score_unpair : 0.f;
  // row: vec[k] = zp[k]; column: vec[x] = zs[x] (the walk of step i reads vec[l+1])
  // (long sequences: the vector does not fit the LDS; it is walked in global memory instead —
  // one workgroup reads what it wrote itself a barrier earlier, through its own CU's L1)
  float* gv = col ? q.zs : q.zp;
  float* vec = use_lds ? lds_vec : gv;
  if (use_lds)
    for (uint32_t x = t; x <= n; x += 1024u) vec[x] = gv[x];
  __syncthreads();
  // operand of step d: row: ZRE[j*ld + k], k = 0..j (j = d); column: QA[i*ld + l], l = i+1..n-1
  auto operand = [&](uint32_t d) -> const float* {
    return col ? q.m[T_QA] + static_cast<size_t>(n - 1u - d) * ld + (n - d)  // l = i+1 at index 0
               : q.m[T_ZRE] + static_cast<size_t>(d) * ld;
  };
  auto terms = [&](uint32_t d) { return col ? d : d + 1u; };
  float pf[4];
  {
    const float* __restrict__ op = operand(dlo);
    const uint32_t len = terms(dlo);
#pragma unroll
    for (int u = 0; u < 4; u++) pf[u] = t + 1024u * u < len ? op[t + 1024u * u] : kNegInf;
  }
  for (uint32_t d = dlo; d <= dtop; d++) {
    const float* __restrict__ op = operand(d);
    const uint32_t len = terms(d);
    const uint32_t i = n - 1u - d;
    // row: term k pairs with vec[k]; column: term idx (l = i+1+idx) with vec[l+1] = vec[i+2+idx]
    const uint32_t vo = col ? i + 2u : 0u;
    float x[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t k = t + 1024u * u;
      x[u] = k < len ? pf[u] + vec[vo + k] : kNegInf;
    }
    Acc a = acc_empty();
    acc_add4(a, x[0], x[1], x[2], x[3]);
    for (uint32_t k = t + 4096u; k < len; k += 1024u) acc_add(a, op[k] + vec[vo + k]);
    if (d < dtop) {  // the next step's operand (final data: independent of this step's result)
      const float* __restrict__ opn = operand(d + 1u);
      const uint32_t lenn = terms(d + 1u);
#pragma unroll
      for (int u = 0; u < 4; u++) pf[u] = t + 1024u * u < lenn ? opn[t + 1024u * u] : kNegInf;
    }
    a = wave_reduce(a);
    if (lane == 0u) {
      red[wave][0] = a.m;
      red[wave][1] = a.s;
    }
    __syncthreads();
    if (wave == 0u) {
      Acc z = lane < 16u ? Acc{red[lane][0], red[lane][1]} : acc_empty();
      z = wave_reduce(z);
      if (lane == 0u) {
        float v;
        if (col) {
          z.m += ext_bp;
          acc_add(z, vec[i + 1u] + ext_un);
          v = acc_value(z);
          vec[i] = v;
          q.zs[i] = v;
        } else {
          acc_add(z, CONTRA ? ext_un * static_cast<float>(d + 1u) : 0.f);
          v = acc_value(z);
          vec[d + 1u] = v;
          q.zp[d + 1u] = v;
        }
      }
    }
    __syncthreads();
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

#include "rnamc_tree_lane.h"
#include "rnamc_tree_mx.h"

#define RNAMC_TREE_LAUNCH(K, C, T, U)                                                            \
  do {                                                                                            \
    const uint32_t gx_ = tree_grid<T>(cells, nd0, nd_count, max_n, nseq, pol, ah);                      \
    hipLaunchKernelGGL((K<C, T, U>), dim3(gx_, nseq, 1), dim3(T < 256 ? 256 : T), 0, st,          \
                       b.workspace + b.one.ws_off, static_cast<uint64_t>(b.one.msz), b.one.n, b.one.ld, d, thr,  \
                       b.use_one, b, two ? 0 : 1, ah);                                            \
  } while (0)
// Threads per cell by the length of a cell's sums (`terms`) and the number of cells: a cell's
// lanes walk its sums four steps per stream at a time, so `terms / (4 * threads)` dependent
// round trips set the launch's duration; more threads per cell while the chip has room for them.
// ... and no more waves than the chip holds at once (the kernels take ~88 VGPRs: 5 waves per
// SIMD, 5120 per chip): a launch whose waves need a second round pays every fixed stage of its
// dependent chain twice (measured: 8192 waves of 256-thread groups took as long as 4096 would
// have taken twice).
// grid of a sweep launch: the workgroups of its own cells, then one wave per row of the next
// launch's first diagonal (the ahead role)
template <int T>
static uint32_t tree_grid(uint32_t cells, uint32_t nd0, uint32_t nd_count, uint32_t max_n, uint32_t nseq_,
                          const TreePolicy& pol, Ahead& ah) {
  constexpr uint32_t per = T < 256 ? 256 / T : 1, block = T < 256 ? 256 : T;
  ah.main_blocks = (cells + per - 1) / per;
  ah.nd0 = nd0;
  ah.nd_count = (nd_count && nd0 < max_n) ? nd_count : 0u;
  uint32_t rows = ah.nd_count ? max_n - nd0 : 0u;
  // one ahead wave per cell while the launch's waves still fit the chip at once
  if (ah.nd_count == 2u && static_cast<uint64_t>(cells + 2u * rows) * nseq_ <= pol.ahead_waves) {
    ah.flags |= 2u;
    rows *= 2u;
  }
  uint32_t ahead_blocks = (rows + block / 64 - 1) / (block / 64);
  // (xcd_chunk: one wave per cell pair, launches of at least 64 workgroups a sequence)
  if (pol.xcd_rows && T == 64 && ah.main_blocks >= 64u) {
    ah.flags |= 4u;
    ah.main_blocks = (ah.main_blocks + 63u) & ~63u;
    ahead_blocks = (ahead_blocks + 63u) & ~63u;
  }
  return ah.main_blocks + ahead_blocks;
}
static int tree_tpc(uint64_t cells, uint32_t terms, int64_t knob, const TreePolicy& pol) {
  if (knob == 64 || knob == 128 || knob == 256 || knob == 1024) return static_cast<int>(knob);
  const uint64_t kWaves = pol.waves;
  if (terms <= pol.short_terms || cells * 2u > kWaves) return 64;
  if (cells * 4u > kWaves) return 128;
  if (terms <= 2048u || cells * 16u > kWaves) return 256;
  return 1024;
}

void launch_tree_static(const TreeBatch& b, bool contra, uint32_t nseq, uint32_t max_n, hipStream_t st) {
  const uint64_t cells = static_cast<uint64_t>(max_n) * (b.lane ? ((max_n + 31u) & ~31u) + 32u : max_n);
  const uint32_t gx = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>((cells + 255) / 256, 4096)));
  if (contra)
    hipLaunchKernelGGL(k_tree_static<true>, dim3(gx, nseq, 1), dim3(256), 0, st, b);
  else
    hipLaunchKernelGGL(k_tree_static<false>, dim3(gx, nseq, 1), dim3(256), 0, st, b);
}

void launch_tree_inside(const TreeBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                        int64_t tpc_knob, bool two, uint32_t thr, bool use_far, uint32_t nd0,
                        uint32_t nd_count, const TreePolicy& pol, hipStream_t st) {
  const uint32_t cells = max_n - d;
  Ahead ah{use_far ? 1u : 0u, 0u, 0u, 0u};
  // (banded: the sums are short whatever d is; what more threads per cell buy is one closing-pair
  // block per wave, so the group is as wide as the chip has room for)
  const int tpc = tree_tpc(static_cast<uint64_t>(cells) * nseq, thr ? 1024u : d, tpc_knob, pol);
  if (use_far) {  // (one wave per cell pair: what wider groups bought was a block per wave)
    if (contra) RNAMC_TREE_LAUNCH(k_tree_inside2, true, 64, true);
    else RNAMC_TREE_LAUNCH(k_tree_inside2, false, 64, true);
  } else if (contra) {
    if (tpc == 64) RNAMC_TREE_LAUNCH(k_tree_inside2, true, 64, false);
    else if (tpc == 128) RNAMC_TREE_LAUNCH(k_tree_inside2, true, 128, false);
    else if (tpc == 256) RNAMC_TREE_LAUNCH(k_tree_inside2, true, 256, false);
    else RNAMC_TREE_LAUNCH(k_tree_inside2, true, 1024, false);
  } else {
    if (tpc == 64) RNAMC_TREE_LAUNCH(k_tree_inside2, false, 64, false);
    else if (tpc == 128) RNAMC_TREE_LAUNCH(k_tree_inside2, false, 128, false);
    else if (tpc == 256) RNAMC_TREE_LAUNCH(k_tree_inside2, false, 256, false);
    else RNAMC_TREE_LAUNCH(k_tree_inside2, false, 1024, false);
  }
}

void launch_tree_outside(const TreeBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                         int64_t tpc_knob, bool two, uint32_t thr, bool use_far, uint32_t nd0,
                         uint32_t nd_count, const TreePolicy& pol, hipStream_t st) {
  const uint32_t cells = max_n - d;
  Ahead ah{use_far ? 1u : 0u, 0u, 0u, 0u};
  const int tpc = tree_tpc(static_cast<uint64_t>(cells) * nseq, thr ? 1024u : max_n - d, tpc_knob, pol);
  if (use_far) {
    if (contra) RNAMC_TREE_LAUNCH(k_tree_outside2, true, 64, true);
    else RNAMC_TREE_LAUNCH(k_tree_outside2, false, 64, true);
  } else if (contra) {
    if (tpc == 64) RNAMC_TREE_LAUNCH(k_tree_outside2, true, 64, false);
    else if (tpc == 128) RNAMC_TREE_LAUNCH(k_tree_outside2, true, 128, false);
    else if (tpc == 256) RNAMC_TREE_LAUNCH(k_tree_outside2, true, 256, false);
    else RNAMC_TREE_LAUNCH(k_tree_outside2, true, 1024, false);
  } else {
    if (tpc == 64) RNAMC_TREE_LAUNCH(k_tree_outside2, false, 64, false);
    else if (tpc == 128) RNAMC_TREE_LAUNCH(k_tree_outside2, false, 128, false);
    else if (tpc == 256) RNAMC_TREE_LAUNCH(k_tree_outside2, false, 256, false);
    else RNAMC_TREE_LAUNCH(k_tree_outside2, false, 1024, false);
  }
#undef RNAMC_TREE_LAUNCH
}

void launch_tree_mid(const TreeBatch& b, bool outside, uint32_t dlo, uint32_t dhi, uint32_t thr,
                     uint32_t max_n, uint32_t nseq, const TreePolicy& pol, hipStream_t st) {
  if (dlo >= max_n || dhi < dlo || nseq == 0) return;
  if (pol.mid_mx) {
    launch_tree_mid_mx(b, outside, dlo, dhi, thr, max_n, nseq, st);
    return;
  }
  const uint32_t tiles_i = (max_n - dlo + kMidTI - 1) / kMidTI;
  const uint32_t tiles_z = ((dhi - dlo) / kMidTD + 1) * (outside ? 2u : 1u);
  const uint32_t wgs = pol.mid_wgs ? pol.mid_wgs : (max_n >= 12288u ? 1024u : (max_n >= 6144u ? 512u : 256u));
  const uint32_t gx = std::max(1u, std::min(tiles_i * tiles_z, (wgs + nseq - 1) / nseq));
  hipLaunchKernelGGL(k_tree_mid, dim3(gx, nseq, 1), dim3(64 * kMidWaves), 0, st, b, dlo, dhi, thr,
                     outside ? 1 : 0, tiles_i, tiles_z);
}

void launch_tree_ext(const TreeBatch& b, bool contra, uint32_t dlo, uint32_t dhi, uint32_t max_n,
                     uint32_t nseq, hipStream_t st) {
  if (dlo >= max_n || dhi < dlo || nseq == 0) return;
  const bool use_lds = max_n <= 12000u;  // (48 KB of dynamic LDS; beyond, the vector stays in global memory)
  const size_t lds = use_lds ? (static_cast<size_t>(max_n) + 2u) * sizeof(float) : 0;
  if (contra)
    hipLaunchKernelGGL(k_tree_ext<true>, dim3(2, nseq, 1), dim3(1024), lds, st, b, dlo, dhi, use_lds ? 1 : 0);
  else
    hipLaunchKernelGGL(k_tree_ext<false>, dim3(2, nseq, 1), dim3(1024), lds, st, b, dlo, dhi, use_lds ? 1 : 0);
}

namespace {
__global__ void k_tree_spin(unsigned long long ticks, unsigned int* sink) {
  // bounded: every wave leaves after `ticks` of the 100-MHz real-time counter (or 2^20 rounds)
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned int rounds = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && rounds < (1u << 20)) {
    __builtin_amdgcn_s_sleep(32);
    rounds++;
  }
  if (sink && rounds == 0xffffffffu) *sink = rounds;
}
}  // namespace

int tree_side_stream_probe(hipStream_t main, hipStream_t side) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int verdict = 0;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    if (e0) (void)hipEventDestroy(e0);
    return 0;
  }
  do {
    if (hipStreamSynchronize(main) != hipSuccess || hipStreamSynchronize(side) != hipSuccess) break;
    constexpr unsigned long long kSpinTicks = 12000;  // 120 us at 100 MHz
    hipLaunchKernelGGL(k_tree_spin, dim3(1), dim3(64), 0, side, kSpinTicks, static_cast<unsigned int*>(nullptr));
    if (hipEventRecord(e0, main) != hipSuccess) break;
    hipLaunchKernelGGL(k_tree_spin, dim3(1), dim3(64), 0, main, 0ull, static_cast<unsigned int*>(nullptr));
    if (hipEventRecord(e1, main) != hipSuccess) break;
    if (hipStreamSynchronize(main) != hipSuccess || hipStreamSynchronize(side) != hipSuccess) break;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) break;
    verdict = ms < 0.06f ? 1 : 2;  // (a trivial kernel behind a free queue: ~10 us)
  } while (false);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return verdict;
}

void launch_tree_finalize(const TreeBatch& b, uint32_t nseq, uint32_t max_n, hipStream_t st) {
  const uint64_t elems = static_cast<uint64_t>(max_n) * (max_n + 1) / 2;
  uint32_t gx = static_cast<uint32_t>(std::min<uint64_t>((elems + 1023) / 1024, 1024));
  if (gx == 0) gx = 1;
  hipLaunchKernelGGL(k_tree_finalize, dim3(gx, nseq, 1), dim3(256), 0, st, b);
}

}  // namespace rnamc
