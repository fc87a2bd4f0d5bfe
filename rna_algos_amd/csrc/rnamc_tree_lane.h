// rnamc_tree_lane.h — tree-order sweeps of a BATCH: a lane per cell (included by rnamc_tree.hip).
//
// The launches of rnamc_tree.hip give a wave to every cell pair: right for a lone long sequence
// (its cost is the chain of launches, and a wave walks a cell's sums in one round trip), wasteful on
// a batch, where a launch holds 10^5 cells and the chip's time goes to ~1 400 wave instructions per
// cell, most of them one cell's scalar recurrences evaluated 64 lanes wide (PMC, round 4: VALU 0.19
// per SIMD-cycle, SALU 0.49 per CU-cycle, five waves per SIMD half of their time waiting — nothing
// saturated, nothing to tune).  Here a wave holds 64 consecutive rows of ONE diagonal, lane = cell,
// as the reference-order kernels do: every recurrence once per lane, the sums as per-lane loops.
// What makes the loops' loads coalesced is the diagonal-major layout (T_*_D, rnamc_device.h): the
// operand of term t of cell (i, i+d) sits at [f(d,t) * ld + i + g(t)] for all of these sums —
//   closing-pair block (src/mccaskill_algo.rs:306-342): enclosed pair (i+1+a, j-1-b), span d-2-a-b
//   sums_multibranch   (344-351): Q1(i, i+x) span x, Zr_mb(i+1+x, j) span d-1-x
//   probs_multibranch  (540-557): W(i, j+1+x) span d+1+x, Q1(j+1, j+x) span x-1
//   L_e cases 1 and 3  (594-601): Q1(i-x, i-1) span x-1, R(i-1-x, j) span d+1+x
//   enclosing 2-loops  (562-593): closing pair (i-1-a, j+1+b), span d+2+a+b
// so 64 lanes read 64 consecutive floats.  The cubic terms stay with k_tree_mid (LDS-tiled, beside the
// sweep), sums_external with k_tree_ext: both read the row- / column-major copies, which these
// kernels keep writing (a scattered 4-byte store per cell and matrix: cheap next to the sums).
// The per-cell statics are read where k_tree_static put them (row-major: six scattered loads a cell).
// One diagonal per launch: a batch's launches are fat, their number is not what costs.
//
// Same arithmetic as the wave-per-cell launches ({max, sum exp2} accumulators, exact two-term
// logsumexp), other grouping of the terms: results agree to rounding, not to the bit (neither is the
// reference's order; tests/test_gpu_tree.py holds both against the f64 value of the recurrences).

namespace {

// A lane's sums are loops of dependent round trips, and a launch of a 256-sequence group is only
// ~4 000 blocks of 64 cells: what sets the pace is how many loads are in flight.  Two measures:
// eight terms a step with every load of the step issued before the first is used, and kLaneParts
// waves to a block of 64 cells — wave p takes the steps p, p + P, p + 2P .. of every sum, the
// partial {max, sum} pairs meet in LDS and the block's first wave finishes the cells.
#ifndef RNAMC_LANE_PARTS
#define RNAMC_LANE_PARTS 4
#endif
constexpr uint32_t kLaneParts = RNAMC_LANE_PARTS;
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void acc_add8(Acc& a, const float (&x)[8]) {
  const float m01 = __builtin_fmaxf(__builtin_fmaxf(x[0], x[1]), x[2]), m23 = __builtin_fmaxf(__builtin_fmaxf(x[3], x[4]), x[5]);
  const float mn = __builtin_fmaxf(__builtin_fmaxf(a.m, m01), __builtin_fmaxf(__builtin_fmaxf(m23, x[6]), x[7]));
  const float mb = mn * kL2E;
  float e[8];
#pragma unroll
  for (int u = 0; u < 8; u++) e[u] = ex2(__builtin_fmaf(x[u], kL2E, -mb));
  const float sum = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
  a.s = __builtin_fmaf(a.s, ex2((a.m - mn) * kL2E), sum);
  a.m = mn;
}

// (+) of A[oa(x)] + B[ob(x)] over this wave's steps of x in [lo, hi), x < mine (the lane's own end)
template <uint32_t PARTS = kLaneParts, typename FA, typename FB>
__device__ __forceinline__ void lane_sum(Acc& acc, const float* __restrict__ A, const float* __restrict__ B, uint32_t lo,
                                         uint32_t hi, uint32_t mine, uint32_t part, FA oa, FB ob) {
  for (uint32_t x = lo + 8u * part; x < hi; x += 8u * PARTS) {
    float va[8], vb[8], v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t xv = x + static_cast<uint32_t>(u);
      const bool ok = xv < hi && xv < mine;  // (a lane past its end reads the streams' first elements)
      va[u] = A[ok ? oa(xv) : size_t{0}];
      vb[u] = B[ok ? ob(xv) : size_t{0}];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t xv = x + static_cast<uint32_t>(u);
      v[u] = (xv < hi && xv < mine) ? va[u] + vb[u] : kNegInf;
    }
    acc_add8(acc, v);
  }
}

// The generic 2-loops of one cell, class by class: this wave's steps of the slots e < cnt of the
// class's list (ordered by a + b), the class's plane at diagonal `dbase -/+ (a + b)`, row
// `i +/- (1 + a)`; st4: the cell's own class scores; own: added to every term (outside: sums_close of
// the cell; there a slot counts for the lanes whose row has room for it, a < i and b < room = n - 1 - j).
// A step's issue slots are what a launch's time goes to (the first form of this loop took ~25 scalar
// and ~14 vector instructions a slot): the plane is a buffer resource, the slot's row a scalar byte
// offset, the lane's part of the address one register for the whole kernel (buffer_load .. offen: no
// vector instruction per load); what is the same for all of a class's terms — the cell's own class
// score — is added once, to the class's accumulator.
template <bool OUTSIDE, bool TAIL>
__device__ __forceinline__ void lane_generic_step(Acc& acc, __amdgpu_buffer_rsrc_t plane, const uint32_t* __restrict__ gs,
                                                  const float* __restrict__ gl, uint32_t e, uint32_t cnt, uint32_t ld,
                                                  uint32_t dbase, uint32_t i, uint32_t voff, uint32_t room) {
  const u32x8 sl8 = *reinterpret_cast<const __attribute__((address_space(4))) u32x8*>(reinterpret_cast<uintptr_t>(gs + e));
  const f32x8 ln8 = *reinterpret_cast<const __attribute__((address_space(4))) f32x8*>(reinterpret_cast<uintptr_t>(gl + e));
  float g[8];
  bool okl[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    // (uniform; past the count the step's first slot stands in: a later slot's diagonal may not exist)
    const uint32_t sl = (!TAIL || e + static_cast<uint32_t>(u) < cnt) ? sl8[u] : sl8[0];
    const uint32_t a = sl & 255u, sab = sl >> 8;
    okl[u] = true;
    if (OUTSIDE) okl[u] = a < i && sab - a < room;
    // (outside, voff = 4 (max(i, 1) - 1): a lane without room reads a neighbour of the row)
    const uint32_t soff = 4u * (OUTSIDE ? (dbase + sab) * ld - a : (dbase - sab) * ld + a);
    g[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(plane, static_cast<int>(voff), static_cast<int>(soff), 0));
  }
  __builtin_amdgcn_sched_barrier(0);
  float x[8];
#pragma unroll
  for (int u = 0; u < 8; u++) x[u] = ((!TAIL || e + static_cast<uint32_t>(u) < cnt) && okl[u]) ? g[u] + ln8[u] : kNegInf;
  acc_add8(acc, x);
}
// two full steps of the wave at once: sixteen loads in flight before the first is used (a launch of the
// batch form holds ~2 300 of these waves on a chip with room for 8 000: its time is the waves' chain of
// round trips, ~15 of them with eight slots a step)
template <bool OUTSIDE>
__device__ __forceinline__ void lane_generic_step2(Acc& acc, __amdgpu_buffer_rsrc_t plane, const uint32_t* __restrict__ gs,
                                                   const float* __restrict__ gl, uint32_t e0, uint32_t e1, uint32_t ld,
                                                   uint32_t dbase, uint32_t i, uint32_t voff, uint32_t room) {
  const u32x8 sa = *reinterpret_cast<const __attribute__((address_space(4))) u32x8*>(reinterpret_cast<uintptr_t>(gs + e0));
  const u32x8 sb = *reinterpret_cast<const __attribute__((address_space(4))) u32x8*>(reinterpret_cast<uintptr_t>(gs + e1));
  const f32x8 la = *reinterpret_cast<const __attribute__((address_space(4))) f32x8*>(reinterpret_cast<uintptr_t>(gl + e0));
  const f32x8 lb = *reinterpret_cast<const __attribute__((address_space(4))) f32x8*>(reinterpret_cast<uintptr_t>(gl + e1));
  float g[16];
  bool okl[16];
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const uint32_t sl = u < 8 ? sa[u & 7] : sb[u & 7];
    const uint32_t a = sl & 255u, sab = sl >> 8;
    okl[u] = true;
    if (OUTSIDE) okl[u] = a < i && sab - a < room;
    const uint32_t soff = 4u * (OUTSIDE ? (dbase + sab) * ld - a : (dbase - sab) * ld + a);
    g[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(plane, static_cast<int>(voff), static_cast<int>(soff), 0));
  }
  __builtin_amdgcn_sched_barrier(0);
  float x[8], y[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    x[u] = okl[u] ? g[u] + la[u] : kNegInf;
    y[u] = okl[u + 8] ? g[u + 8] + lb[u] : kNegInf;
  }
  acc_add8(acc, x);
  acc_add8(acc, y);
}
// Class 3 in runs (TreeTabs::g4slot): one step = four groups = sixteen slots off FOUR 16-byte loads and
// four scalar offsets (the slot-by-slot form: sixteen loads, sixteen offsets at ~11 scalar instructions
// each — the scalar unit, not the vector ALU, was the busiest part of these kernels by the SQ counters).
// Inside the run of a level lies at rows i+1+a0 .. of diagonal dbase - s; outside at rows i-1-a0 .. DOWN
// of diagonal dbase + s, so the load starts three floats lower and the elements come reversed.
typedef float f32x4b __attribute__((ext_vector_type(4)));
template <bool OUTSIDE>
__device__ __forceinline__ void lane_generic_run4(Acc& acc, __amdgpu_buffer_rsrc_t plane, const uint32_t* __restrict__ gs,
                                                  const float (*__restrict__ gl)[4], uint32_t g, uint32_t ng, uint32_t ld,
                                                  uint32_t dbase, uint32_t i, uint32_t voff, uint32_t room) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 sl = *reinterpret_cast<const __attribute__((address_space(4))) u32x4*>(reinterpret_cast<uintptr_t>(gs + g));
  f32x4b v[4];
  float ln[16];
  bool okl[16];
#pragma unroll
  for (int x = 0; x < 4; x++) {
    const uint32_t a0 = sl[x] & 255u, sab = sl[x] >> 8;
    const uint32_t soff = 4u * (OUTSIDE ? (dbase + sab) * ld - a0 - 3u : (dbase - sab) * ld + a0);
    v[x] = __builtin_bit_cast(f32x4b, __builtin_amdgcn_raw_buffer_load_b128(plane, static_cast<int>(voff), static_cast<int>(soff), 0));
    const f32x4b l4 = *reinterpret_cast<const __attribute__((address_space(4))) f32x4b*>(reinterpret_cast<uintptr_t>(&gl[g + x][0]));
    const bool in = g + static_cast<uint32_t>(x) < ng;  // (uniform: a group past the cell's span counts for nothing)
#pragma unroll
    for (int u = 0; u < 4; u++) {
      ln[4 * x + u] = in ? l4[u] : kNegInf;
      okl[4 * x + u] = true;
      if (OUTSIDE) okl[4 * x + u] = a0 + static_cast<uint32_t>(u) < i && sab - a0 - static_cast<uint32_t>(u) < room;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  float xa[8], xb[8];
#pragma unroll
  for (int x = 0; x < 4; x++)
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const float val = OUTSIDE ? v[x][3 - u] : v[x][u];
      const float t = okl[4 * x + u] ? val + ln[4 * x + u] : kNegInf;
      if (x < 2) xa[4 * x + u] = t; else xb[4 * (x - 2) + u] = t;
    }
  acc_add8(acc, xa);
  acc_add8(acc, xb);
}
// Classes 0 and 1 by level (TreeTabs::elen): step t = the levels 2 + 2 t and 3 + 2 t, four loads each at the
// level's row plus 0, s (plane 0: bulges) and 1, s - 1 (plane 1: 1 x many); c0 / c1: the cell's own class scores.
template <bool OUTSIDE>
__device__ __forceinline__ void lane_generic_edge(Acc& acc, __amdgpu_buffer_rsrc_t p0, __amdgpu_buffer_rsrc_t p1,
                                                  const float (*__restrict__ el)[4], uint32_t t, uint32_t smax, uint32_t ld,
                                                  uint32_t dbase, uint32_t i, uint32_t voff, uint32_t room, float c0,
                                                  float c1) {
  float g[8], ln[8];
  bool okl[8];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const uint32_t s = 2u + 2u * t + static_cast<uint32_t>(h);  // (uniform)
    const bool in = s <= smax;
    const uint32_t sc = in ? s : 2u;  // (a level past the cell's span: the first level's addresses, nothing counted)
    const f32x4b l4 = *reinterpret_cast<const __attribute__((address_space(4))) f32x4b*>(reinterpret_cast<uintptr_t>(&el[sc][0]));
    const uint32_t row4 = 4u * (OUTSIDE ? dbase + sc : dbase - sc) * ld;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t a = u == 0 ? 0u : (u == 1 ? sc : (u == 2 ? 1u : sc - 1u));
      const uint32_t soff = OUTSIDE ? row4 - 4u * a : row4 + 4u * a;
      g[4 * h + u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(u < 2 ? p0 : p1, static_cast<int>(voff), static_cast<int>(soff), 0));
      ln[4 * h + u] = (in ? l4[u] : kNegInf) + (u < 2 ? c0 : c1);
      okl[4 * h + u] = true;
      if (OUTSIDE) okl[4 * h + u] = a < i && sc - a < room;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; k++) x[k] = okl[k] ? g[k] + ln[k] : kNegInf;
  acc_add8(acc, x);
}
template <bool CONTRA, bool OUTSIDE>
__device__ __forceinline__ void lane_generic(Acc& acc, const TreeBatch& b, float* x4, size_t msz, uint32_t ld,
                                             uint32_t smax, uint32_t dbase, uint32_t i, const float4& st4, float own,
                                             uint32_t room, uint32_t part) {
  const uint32_t voff = 4u * (OUTSIDE ? max(i, 1u) - 1u : i + 1u);
  uint32_t turn = part;  // (the classes' steps are dealt to the block's waves in one round-robin)
  {
    // class 3 first, in runs of four slots (its own round-robin: steps of four groups)
    const uint32_t ng = sload(&b.tabs->g4count[CONTRA ? 1 : 0][smax]);
    const __amdgpu_buffer_rsrc_t plane =
        __builtin_amdgcn_make_buffer_rsrc(x4 + 3u * msz, 0, static_cast<int>(msz * sizeof(float)), 0x00020000);
    Acc ac = acc_empty();
    for (uint32_t g = 4u * part; g < ng; g += 4u * kLaneParts)
      lane_generic_run4<OUTSIDE>(ac, plane, b.tabs->g4slot[CONTRA ? 1 : 0], b.tabs->g4len[CONTRA ? 1 : 0], g, ng, ld, dbase, i,
                                 voff, room);
    ac.m += OUTSIDE ? own + st4.w : st4.w;  // (an empty accumulator stays empty: s = 0)
    acc_merge(acc, ac);
  }
  {
    // classes 0 and 1 by level
    const __amdgpu_buffer_rsrc_t p0 = __builtin_amdgcn_make_buffer_rsrc(x4, 0, static_cast<int>(msz * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t p1 =
        __builtin_amdgcn_make_buffer_rsrc(x4 + msz, 0, static_cast<int>(msz * sizeof(float)), 0x00020000);
    const float c0 = OUTSIDE ? own + st4.x : st4.x, c1 = OUTSIDE ? own + st4.y : st4.y;
    Acc ac = acc_empty();
    for (uint32_t t = part; t < smax / 2u; t += kLaneParts)  // (levels 2 .. smax, two a step)
      lane_generic_edge<OUTSIDE>(ac, p0, p1, b.tabs->elen[CONTRA ? 1 : 0], t, smax, ld, dbase, i, voff, room, c0, c1);
    acc_merge(acc, ac);
  }
#pragma unroll
  for (uint32_t c = 2u; c < 3u; c++) {
    const uint32_t start = sload(&b.tabs->gstart[CONTRA ? 1 : 0][c]);
    const uint32_t cnt = sload(&b.tabs->gcount[CONTRA ? 1 : 0][c][smax]);
    const uint32_t* __restrict__ gs = b.tabs->gslot[CONTRA ? 1 : 0] + start;
    const float* __restrict__ gl = b.tabs->glen[CONTRA ? 1 : 0] + start;
    // (raw buffer over the plane: byte offsets, reads past the end return 0 instead of faulting)
    const __amdgpu_buffer_rsrc_t plane =
        __builtin_amdgcn_make_buffer_rsrc(x4 + c * msz, 0, static_cast<int>(msz * sizeof(float)), 0x00020000);
    const uint32_t steps = (cnt + 7u) >> 3, full = cnt >> 3;
    Acc ac = acc_empty();
    uint32_t st = turn;
    for (; st + kLaneParts < full; st += 2u * kLaneParts)
      lane_generic_step2<OUTSIDE>(ac, plane, gs, gl, 8u * st, 8u * (st + kLaneParts), ld, dbase, i, voff, room);
    for (; st < full; st += kLaneParts) lane_generic_step<OUTSIDE, false>(ac, plane, gs, gl, 8u * st, cnt, ld, dbase, i, voff, room);
    if (st < steps) lane_generic_step<OUTSIDE, true>(ac, plane, gs, gl, 8u * st, cnt, ld, dbase, i, voff, room);
    turn = (turn + kLaneParts - steps % kLaneParts) % kLaneParts;
    ac.m += OUTSIDE ? own + pick(st4, c) : pick(st4, c);  // (an empty accumulator stays empty: s = 0)
    acc_merge(acc, ac);
  }
}

// ---- the cells of a diagonal that may pair, in row order (k_tlane_list, once per group): the 2-loop sums
// run over THESE — 6 of 16 base combinations are canonical, so a wave of 64 consecutive rows would walk
// its ~490 slots with 24 lanes at work.  list[d * ld + c] = row of the c-th such cell of diagonal d, the
// count in the row's last float (ld >= n + 32: never a list entry).  A wave then holds 64 LISTED cells
// (their rows span ~170 floats: still whole lines) and every slot stays wave-uniform.
__device__ __forceinline__ const uint32_t* lane_list(const TSeq& q, uint32_t d) {
  return reinterpret_cast<const uint32_t*>(q.m[T_LIST]) + static_cast<size_t>(d) * q.ld;
}
__global__ void __launch_bounds__(256) k_tlane_list(TreeBatch b) {
  __shared__ uint32_t wsum[4];
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld, d = blockIdx.x;
  if (d >= n) return;
  uint32_t* __restrict__ out = reinterpret_cast<uint32_t*>(q.m[T_LIST]) + static_cast<size_t>(d) * ld;
  const float* __restrict__ mbc = q.m[T_MBC] + static_cast<size_t>(d) * ld;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t base = 0;
  for (uint32_t i0 = 0; i0 < n - d; i0 += 256u) {
    const uint32_t i = i0 + threadIdx.x;
    const bool on = i < n - d && mbc[i] > kNegInf;
    const uint64_t bal = __builtin_amdgcn_ballot_w64(on);
    const uint32_t before = static_cast<uint32_t>(__builtin_popcountll(bal & ((1ull << lane) - 1ull)));
    if (lane == 0u) wsum[wave] = static_cast<uint32_t>(__builtin_popcountll(bal));
    __syncthreads();
    uint32_t off = base;
    for (uint32_t w = 0; w < wave; w++) off += wsum[w];
    if (on) out[off + before] = i;
    base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0u) out[ld - 1u] = base;
}

// ---- inside sweep (src/mccaskill_algo.rs:296-351 / 430-486).  Two roles in one launch:
//   blockIdx.x <  nb_a : the cells of diagonal d, a lane per row: sums_multibranch's in-band terms and the
//                        row recurrences (the cell's sums_accessible was written a launch earlier);
//   blockIdx.x >= nb_a : the closing-pair blocks of diagonal d_b = d + 1, a lane per LISTED cell: a 2-loop
//                        (i,j) -> (i+1+a, j-1-b) and Qm(i+1, j-1) read diagonals <= d_b - 2, i.e. nothing
//                        this launch writes.  (d_b >= n: no such role; nb_a = 0: that role alone.)
template <bool CONTRA>
__global__ void __launch_bounds__(64 * kLaneParts) k_tlane_inside(TreeBatch b, uint32_t d, uint32_t thr, uint32_t nb_a,
                                                                 uint32_t d_b) {
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t part = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)));
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 4) return;
#endif
  const size_t msz = static_cast<size_t>(q.m[1] - q.m[0]);
  if (blockIdx.x >= nb_a) {
    // ---- closing-pair blocks of the listed cells of diagonal d_b: hairpin, multibranch term, the explicit
    // small 2-loops; the generic 2-loops' sum was formed by k_tlane_gen (T_GEN_D).  A lane per listed cell.
    const uint32_t db = d_b;
    if (db >= n) return;
    const uint32_t* __restrict__ list = lane_list(q, db);
    const uint32_t cnt = sload(list + (ld - 1u));
    const uint32_t c = (blockIdx.x - nb_a) * (64u * kLaneParts) + threadIdx.x;
    if (c >= cnt) return;
    const uint32_t i = list[c];
    const size_t dg = static_cast<size_t>(db) * ld + i;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    Acc acc = acc_empty();
    const float mbc = q.m[T_MBC][dg];  // (finite: a listed cell)
    const float hp = q.m[T_HP][dg];
    const float4 n4 = reinterpret_cast<const float4*>(q.m[T_NEAR4])[dg];
    const float4 n8 = CONTRA ? zero4 : reinterpret_cast<const float4*>(q.m[T_NEAR8])[dg];
    const float qm = db >= 2u ? q.m[T_QM][dg - 2u * static_cast<size_t>(ld) + 1u] : kNegInf;  // Qm(i+1, j-1)
    Acc gen = acc_empty();
#ifdef RNAMC_DEBUG_KNOBS
    if (!(b.debug & 1))
#endif
    if (db >= 5u) {  // (a generic slot has a + b >= 2)
      const float2 g2 = reinterpret_cast<const float2*>(q.m[T_GEN_D])[dg];
      gen = Acc{g2.x, g2.y};
    }
    const float nr[8] = {n4.x, n4.y, n4.z, n4.w, n8.x, n8.y, n8.z, 0.f};
    float xs[8];
#pragma unroll
    for (uint32_t t = 0; t < 8u; t++) {
      xs[t] = kNegInf;
      if (t < Special<CONTRA>::N) {
        uint32_t a, bb;
        Special<CONTRA>::slot(t, a, bb);
        if (a + bb + 3u <= db) xs[t] = q.m[T_QB_D][static_cast<size_t>(db - 2u - a - bb) * ld + (i + 1u + a)] + nr[t];
      }
    }
    acc_add4(acc, hp, qm + mbc, xs[0], xs[1]);
    if (CONTRA) {
      acc_add2(acc, xs[2], xs[3]);
    } else {
      acc_add4(acc, xs[2], xs[3], xs[4], xs[5]);
      acc_add(acc, xs[6]);
    }
    acc_merge(acc, gen);
    const float qb = acc_value(acc);
    if (qb > kNegInf) {
      const float4 in4 = reinterpret_cast<const float4*>(q.m[T_IN4])[dg];
      q.m[T_QB_D][dg] = qb;
      q.m[T_QA_D][dg] = qb + q.m[T_ACCS][dg];
      float* __restrict__ x4 = q.m[T_X4];
      x4[dg] = qb + in4.x;
      x4[msz + dg] = qb + in4.y;
      x4[2u * msz + dg] = qb + in4.z;
      x4[3u * msz + dg] = qb + in4.w;
    }
    return;
  }
  // ---- the cells of diagonal d: a wave per 64 rows, each lane its whole sums (with the mid-field in front of
  // the band they are at most 2 x band terms: splitting them over four waves and joining through LDS cost more
  // wave launches than it saved chain)
  const uint32_t i = (blockIdx.x * kLaneParts + part) * 64u + lane;
  if ((blockIdx.x * kLaneParts + part) * 64u + d >= n) return;  // (the whole wave; this role has no barrier)
  const bool valid = i + d < n;
  const size_t dg = static_cast<size_t>(d) * ld + i;
  Acc acc[1] = {acc_empty()};
  // sums_multibranch: x = Q1's span, Q1(i, i+x) + Zr_mb(i+1+x, j); banded (thr != 0): the terms with
  // both spans below thr are k_tree_mid's
#ifdef RNAMC_DEBUG_KNOBS
  if (!(b.debug & 2))
#endif
  if (d >= 2u) {
    const float* __restrict__ A = q.m[T_Q1_D] + i;
    const float* __restrict__ B = q.m[T_ZRM_D] + (i + 1u);
    auto oa = [&](uint32_t x) { return static_cast<size_t>(x) * ld; };
    auto ob = [&](uint32_t x) { return static_cast<size_t>(d - 1u - x) * ld + x; };
    const uint32_t mine = valid ? n : 0u;
    if (thr != 0u) {
      lane_sum<1>(acc[0], A, B, 0u, d - thr, mine, 0u, oa, ob);
      lane_sum<1>(acc[0], A, B, thr, d - 1u, mine, 0u, oa, ob);
    } else {
      lane_sum<1>(acc[0], A, B, 0u, d - 1u, mine, 0u, oa, ob);
    }
  }
  if (!valid) return;
  if (thr != 0u && d >= 2u) {
    const float2 mm = q.mid[static_cast<size_t>(d % b.ring) * q.vec + i];
    acc_merge(acc[0], Acc{mm.x, mm.y});
  }
  const float qa = q.m[T_QA_D][dg];  // (-inf unless the cell's closing-pair block was finite)
  const float ext_bp = CONTRA ? b.params->contra.external_score_basepair : 0.f;
  const float ext_un = CONTRA ? b.params->contra.external_score_unpair : 0.f;
  const float mb_bp = CONTRA ? b.params->contra.multibranch_score_basepair : b.params->turner.coeff_num_branches;
  const float mb_un = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  // rightmost-pair sums along the row, their column prefix
  const float zr_e_prev = d >= 1u ? q.m[T_ZRE_D][dg - ld] : kNegInf;              // Zr_ext(i, j-1)
  const float zr_m_prev = (CONTRA && d >= 1u) ? q.m[T_ZRM_D][dg - ld] : kNegInf;  // Zr_mb(i, j-1)
  const float u_next = d >= 1u ? q.m[T_U][dg - ld + 1u] : kNegInf;                // U(i+1, j)
  const float zr_e = lse2(zr_e_prev + ext_un, qa + ext_bp);
  const float zr_m = CONTRA ? lse2(zr_m_prev + mb_un, qa + mb_bp) : zr_e + mb_bp;
  const float u = lse2(u_next + mb_un, zr_m);
  const float qmv = acc_value(acc[0]);
  const float q1 = lse2(u, qmv);
  q.m[T_ZRE_D][dg] = zr_e;
  q.m[T_ZRM_D][dg] = zr_m;
  q.m[T_U][dg] = u;
  q.m[T_QM][dg] = qmv;
  q.m[T_Q1_D][dg] = q1;
}

// ---- outside sweep, from the top (src/mccaskill_algo.rs:528-606 / 640-720).  In this sweep
// T_QM holds probs_multibranch2 and T_U the column prefix of probs_multibranch DIAGONAL-major (their
// only readers are this kernel's neighbours), T_X4 the planes PX4 = (log bpp - sums_close) + CS4[class],
// T_ZRM_D R = Pm (+) Pm2, T_W_D W = (log bpp + mbclose) - sums_close; W row-major (T_ZRE) and R
// column-major (T_ZRM) are kept for k_tree_mid.  Two roles as in the inside sweep:
//   blockIdx.x <  nb_a : the cells of diagonal d (a lane per row): probs_multibranch, L_e, the recurrences,
//                        the pair's probability — its 2-loop part (T_P2_D) was written a launch earlier;
//   blockIdx.x >= nb_a : the enclosing 2-loops (562-593) of the listed cells of diagonal d_b = d - 1: they
//                        read log bpp and PX4 of diagonals >= d_b + 2.  (d_b >= n: no such role.)
template <bool CONTRA>
__global__ void __launch_bounds__(64 * kLaneParts) k_tlane_outside(TreeBatch b, uint32_t d, uint32_t thr, uint32_t nb_a,
                                                                  uint32_t d_b) {
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t part = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)));
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 4) return;
#endif
  const size_t msz = static_cast<size_t>(q.m[1] - q.m[0]);
  if (blockIdx.x >= nb_a) {
    // (the explicit small 2-loops around the listed cells of diagonal d_b: they read diagonals d_b + 2 ..
    // d_b + 6, the most recent ones; the generic ones were summed by k_tlane_gen, T_GEN_D)
    const uint32_t db = d_b;
    if (db >= n) return;
    const uint32_t* __restrict__ list = lane_list(q, db);
    const uint32_t cnt = sload(list + (ld - 1u));
    const uint32_t c = (blockIdx.x - nb_a) * (64u * kLaneParts) + threadIdx.x;
    if (c >= cnt) return;
    const uint32_t i = list[c];
    const uint32_t j = i + db;
    const size_t dg = static_cast<size_t>(db) * ld + i;
    const uint32_t room = n - 1u - j;  // bases right of j
    const float qb = q.m[T_QB_D][dg];
    Acc acc = acc_empty();
    if (qb > kNegInf) {
      float xs[8];
#pragma unroll
      for (uint32_t t = 0; t < 8u; t++) {
        xs[t] = kNegInf;
        if (t < Special<CONTRA>::N) {
          uint32_t a, bb;
          Special<CONTRA>::slot(t, a, bb);
          if (a < i && bb < room) {
            const uint32_t k = i - 1u - a, dd = db + 2u + a + bb;
            const float nqb = q.m[T_QB_D][static_cast<size_t>(dd) * ld + k];
            const float npk = q.out[tri_off(n, dd) + k];
            const float nsc = (t < 4u ? q.m[T_NEAR4] : q.m[T_NEAR8])[4u * (static_cast<size_t>(dd) * ld + k) + (t & 3u)];
            if (nqb > kNegInf) xs[t] = ((npk + qb) - nqb) + nsc;
          }
        }
      }
      acc_add4(acc, xs[0], xs[1], xs[2], xs[3]);
      if (!CONTRA) acc_add4(acc, xs[4], xs[5], xs[6], kNegInf);
    }
    reinterpret_cast<float2*>(q.m[T_P2_D])[dg] = make_float2(acc.m, acc.s);
    return;
  }
  const uint32_t i = (blockIdx.x * kLaneParts + part) * 64u + lane;
  if ((blockIdx.x * kLaneParts + part) * 64u + d >= n) return;  // (the whole wave; this role has no barrier)
  const bool valid = i + d < n;
  const uint32_t j = i + d;
  const size_t dg = static_cast<size_t>(d) * ld + i, dg1 = dg + ld;  // (i, j) and (i, j+1)
  const float mb_un = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  const float ext_bp = CONTRA ? b.params->contra.external_score_basepair : 0.f;
  const float abr = CONTRA ? b.params->contra.multibranch_score_basepair : b.params->turner.coeff_num_branches;
  const uint32_t room = valid ? n - 1u - j : 0u;  // bases right of j
  float* __restrict__ w_d = q.m[T_W_D];
  float* __restrict__ r_d = q.m[T_ZRM_D];
  float* __restrict__ pm2_d = q.m[T_QM];
  float* __restrict__ sp_d = q.m[T_U];
  const float* __restrict__ q1_d = q.m[T_Q1_D];
  const float qb = valid ? q.m[T_QB_D][dg] : kNegInf;
  const bool paired = qb > kNegInf;

  // [0] probs_multibranch, [1] L_e cases one and three
  Acc acc[2] = {acc_empty(), acc_empty()};
  // probs_multibranch(i,j) (540-543): x = 1 .., W(i, j+1+x) + Q1(j+1, j+x); banded: W's span d+1+x < thr
  const uint32_t hi = thr != 0u ? (thr > d + 1u ? thr - 1u - d : 0u) : n - d;  // (uniform end; room, i < n - d)
#ifdef RNAMC_DEBUG_KNOBS
  if (!(b.debug & 2))
#endif
  if (hi > 1u) {
    auto oa = [&](uint32_t x) { return static_cast<size_t>(d + 1u + x) * ld; };
    auto ob = [&](uint32_t x) { return static_cast<size_t>(x - 1u) * ld + d + 1u; };
    lane_sum<1>(acc[0], w_d + i, q1_d + i, 1u, hi, room, 0u, oa, ob);
  }
  // L_e cases one and three (594-601): x = 1 .., Q1(i-x, i-1) + R(i-1-x, j); banded: R's span d+1+x < thr
#ifdef RNAMC_DEBUG_KNOBS
  if (!(b.debug & 2))
#endif
  if (hi > 1u) {
    auto oa = [&](uint32_t x) { return static_cast<size_t>(x - 1u) * ld - x; };
    auto ob = [&](uint32_t x) { return static_cast<size_t>(d + 1u + x) * ld - 1u - x; };
    lane_sum<1>(acc[1], q1_d + i, r_d + i, 1u, hi, paired ? i : 0u, 0u, oa, ob);
  }
  if (!valid) return;
  if (thr != 0u) {
    const float2 m1 = q.mid[(static_cast<size_t>(b.ring) + d % b.ring) * q.vec + i];
    acc_merge(acc[0], Acc{m1.x, m1.y});
    const float2 m2 = q.mid[(2u * static_cast<size_t>(b.ring) + d % b.ring) * q.vec + i];
    acc_merge(acc[1], Acc{m2.x, m2.y});
  }
  // probs_multibranch2(i,j) from the right neighbour (544-549), R, the column prefix
  const float pm2_next = room >= 1u ? pm2_d[dg1] : kNegInf, w_next = room >= 1u ? w_d[dg1] : kNegInf;
  const float pm2 = lse2(pm2_next + mb_un, w_next);
  const float pm = acc_value(acc[0]);
  const float r = lse2(pm, pm2);
  const float sp_prev = i >= 1u ? sp_d[dg1 - 1u] : kNegInf;  // prefix of column j up to row i-1: cell (i-1, j)
  pm2_d[dg] = pm2;
  r_d[dg] = r;
  sp_d[dg] = lse2(sp_prev + mb_un, pm);
  if (!paired) return;
  // the pair's probability (562-604): external, enclosing 2-loops, multibranch cases
  const float qa = qb + q.m[T_ACCS][dg];
  const float zpi = q.zp[i], zsj = q.zs[j + 1u], ztot = sload(q.zp + n);
  const float2 p2 = reinterpret_cast<const float2*>(q.m[T_P2_D])[dg];
  Acc pa = Acc{p2.x, p2.y};
#ifdef RNAMC_DEBUG_KNOBS
  if (!(b.debug & 1))
#endif
  if (n >= d + 5u) {  // (the generic enclosing 2-loops: k_tlane_gen)
    const float2 g2 = reinterpret_cast<const float2*>(q.m[T_GEN_D])[dg];
    acc_merge(pa, Acc{g2.x, g2.y});
  }
  acc_add(pa, CONTRA ? (((zpi + zsj) + qa) + ext_bp) - ztot : ((zpi + qa) + zsj) - ztot);
  const float A = qa + abr;
  acc_add(pa, A + acc_value(acc[1]));
  acc_add(pa, A + sp_prev);
  const float lp = acc_value(pa);
  if (lp > kNegInf) {
    const float w = (lp + q.m[T_MBC][dg]) - qb;
    const float4 cs = reinterpret_cast<const float4*>(q.m[T_CS4])[dg];
    const float pq = lp - qb;
    q.out[tri_off(n, d) + i] = lp;
    w_d[dg] = w;
    float* __restrict__ x4 = q.m[T_X4];
    x4[dg] = pq + cs.x;
    x4[msz + dg] = pq + cs.y;
    x4[2u * msz + dg] = pq + cs.z;
    x4[3u * msz + dg] = pq + cs.w;
  }
}

// ---- the generic 2-loops (a + b >= 2: ~490 slots a cell, the bulk of the sweeps' loads) of the listed cells
// of kGenDiags CONSECUTIVE diagonals in one launch.  A cell's slots read the 29 diagonals from four below
// (inside) / above (outside) its own, and so do its neighbours' on the next diagonal, one row of slots
// further: a launch per diagonal fetched every line once per diagonal — 12 % L2 hits, the group's window
// is ~70 MB a launch — i.e. each ~29 times over the sweep.  Here the workgroup's twelve waves are three
// diagonals x four parts over the same stretch of the lists (the same rows, give or take the lists'
// drift), in step through the slots ordered by a + b: the lines meet in the CU's L1.  Possible because
// the generic slots never touch the last three diagonals: after a finished diagonal d the sums of d + 1 ..
// d + 4 could go at once.  Three are taken (twelve waves): four — sixteen waves, 1 024 threads — were
// slower (685 against 659 ms on the 1 000-sequence slice).  (The four-diagonal form first gave differing
// results under Turner tables: the sweep's very first batch was enqueued in FRONT of the first diagonal's
// closing-pair blocks and read their X4 before it was written — found with a mid-sweep dump of T_GEN_D,
// scripts/gen4_probe2.py, and fixed in the host schedule; three diagonals never reached that diagonal.)
// The sum of a cell goes to T_GEN_D as a {max, sum} pair; the just-in-time rest of the block — hairpin,
// multibranch, explicit small loops — stays with the sweep's launches.
#ifndef RNAMC_GEN_DIAGS
#define RNAMC_GEN_DIAGS 3
#endif
constexpr uint32_t kGenDiags = RNAMC_GEN_DIAGS;
template <bool CONTRA, bool OUTSIDE>
__global__ void __launch_bounds__(64 * kLaneParts * kGenDiags) k_tlane_gen(TreeBatch b, uint32_t g0, uint32_t gcount) {
  __shared__ float2 red[kGenDiags][kLaneParts][64];
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)));
  const uint32_t part = wave % kLaneParts, ds = wave / kLaneParts;
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 5) return;
#endif
  const size_t msz = static_cast<size_t>(q.m[1] - q.m[0]);
  // this wave's diagonal (inside: g0 + ds, outside: g0 - ds) and whether it has generic slots at all
  const uint32_t e = OUTSIDE ? g0 - ds : g0 + ds;
  const bool has = ds < gcount && (OUTSIDE ? (ds <= g0 && n >= e + 5u) : (e < n && e >= 5u));
  uint32_t cnt = 0;
  const uint32_t* __restrict__ list = nullptr;
  if (has) {
    list = lane_list(q, e);
    cnt = sload(list + (ld - 1u));
  }
  // The grid holds HALF the workgroups a diagonal of listed cells only could need (6 of 16 base combinations
  // pair: a quarter of them found work, the others cost their launch): a workgroup walks its stretches of
  // 64 list entries until the longest of its diagonals' lists ends (uniform: the counts travel through LDS).
  __shared__ uint32_t cmax[kGenDiags];
  if (part == 0u && lane == 0u) cmax[ds] = cnt;
  __syncthreads();
  uint32_t most = 0;
#pragma unroll
  for (uint32_t x = 0; x < kGenDiags; x++) most = max(most, cmax[x]);
  for (uint32_t c0 = blockIdx.x * 64u; c0 < most; c0 += gridDim.x * 64u) {
    const bool live = has && c0 < cnt;
    Acc acc = acc_empty();
    uint32_t i = 0;
    bool valid = false;
    size_t dg = 0;
    if (live) {
      valid = c0 + lane < cnt;
      i = valid ? list[c0 + lane] : list[c0];
      dg = static_cast<size_t>(e) * ld + i;
      if (!OUTSIDE) {
        const float4 cs = reinterpret_cast<const float4*>(q.m[T_CS4])[dg];
        lane_generic<CONTRA, false>(acc, b, q.m[T_X4], msz, ld, min(e - 3u, 30u), e - 2u, i, cs, 0.f, 0u, part);
      } else {
        const float qb = valid ? q.m[T_QB_D][dg] : kNegInf;
        if (qb > kNegInf) {
          const float4 in4 = reinterpret_cast<const float4*>(q.m[T_IN4])[dg];
          lane_generic<CONTRA, true>(acc, b, q.m[T_X4], msz, ld, min(n - 3u - e, 30u), e + 2u, i, in4, qb, n - 1u - (i + e), part);
        }
      }
    }
    if (part != 0u) red[ds][part][lane] = make_float2(acc.m, acc.s);
    __syncthreads();
    if (part == 0u && live && valid) {
#pragma unroll
      for (uint32_t p2 = 1; p2 < kLaneParts; p2++) {
        const float2 v = red[ds][p2][lane];
        acc_merge(acc, Acc{v.x, v.y});
      }
      reinterpret_cast<float2*>(q.m[T_GEN_D])[dg] = make_float2(acc.m, acc.s);
    }
    __syncthreads();  // (the next stretch's partial sums overwrite the exchange area)
  }
}

// ---- a finished band of diagonals [dlo, dhi] from the diagonal-major matrices into the row- and
// column-major ones k_tree_mid and k_tree_ext read (tiles of 64 diagonals x 64 rows or columns through
// LDS: both sides of the copy are 256-byte wave accesses).  blockIdx.z = which copy.
//   inside : Q1 -> T_Q1R (row) and T_Q1C (column), Zr_mb -> T_ZRM (column), Zr_ext -> T_ZRE (column),
//            sums_accessible -> T_QA (row)
//   outside: W -> T_ZRE (row), R -> T_ZRM (column)
__global__ void __launch_bounds__(256) k_tlane_spread(TreeBatch b, uint32_t dlo, uint32_t dhi, int outside) {
  __shared__ float tile[64][65];
  const TSeq q = load_tseq(b, blockIdx.y);
  const int n = static_cast<int>(q.n);
  const uint32_t ld = q.ld;
  const int x0 = static_cast<int>(blockIdx.x) * 64;
  if (x0 >= n) return;
  const float* __restrict__ src;
  float* __restrict__ dst;
  bool colm;
  if (outside) {
    src = blockIdx.z == 0u ? q.m[T_W_D] : q.m[T_ZRM_D];
    dst = blockIdx.z == 0u ? q.m[T_ZRE] : q.m[T_ZRM];
    colm = blockIdx.z != 0u;
  } else {
    const uint32_t z = blockIdx.z;
    src = z <= 1u ? q.m[T_Q1_D] : (z == 2u ? q.m[T_ZRM_D] : (z == 3u ? q.m[T_ZRE_D] : q.m[T_QA_D]));
    dst = z == 0u ? q.m[T_Q1R] : (z == 1u ? q.m[T_Q1C] : (z == 2u ? q.m[T_ZRM] : (z == 3u ? q.m[T_ZRE] : q.m[T_QA])));
    colm = z >= 1u && z <= 3u;
  }
  const int lane = static_cast<int>(threadIdx.x & 63u), wave = static_cast<int>(threadIdx.x >> 6);
  for (int d0 = static_cast<int>(dlo); d0 <= static_cast<int>(dhi); d0 += 64) {
    const int nd = min(64, static_cast<int>(dhi) - d0 + 1);
    // row form: x = row i, the cell (i, i + d); column form: x = column j, the cell (j - d, j)
    for (int dd = wave; dd < nd; dd += 4) {
      const int d = d0 + dd, x = x0 + lane;
      const int i = colm ? x - d : x;
      tile[dd][lane] = (i >= 0 && i + d < n) ? src[static_cast<size_t>(d) * ld + i] : kNegInf;
    }
    __syncthreads();
    for (int xx = wave; xx < 64; xx += 4) {
      const int x = x0 + xx, d = d0 + lane;
      if (lane < nd && x < n) {
        const int i = colm ? x - d : x, j = colm ? x : x + d;
        if (i >= 0 && j < n) dst[colm ? static_cast<size_t>(j) * ld + i : static_cast<size_t>(i) * ld + j] = tile[lane][xx];
      }
    }
    __syncthreads();
  }
}

}  // namespace

void launch_tlane_spread(const TreeBatch& b, bool outside, uint32_t dlo, uint32_t dhi, uint32_t max_n, uint32_t nseq,
                         hipStream_t st) {
  if (dlo > dhi || nseq == 0u) return;
  hipLaunchKernelGGL(k_tlane_spread, dim3((max_n + 63u) / 64u, nseq, outside ? 2u : 5u), dim3(256), 0, st, b, dlo, dhi,
                     outside ? 1 : 0);
}

void launch_tlane_list(const TreeBatch& b, uint32_t max_n, uint32_t nseq, hipStream_t st) {
  if (nseq == 0u || max_n == 0u) return;
  hipLaunchKernelGGL(k_tlane_list, dim3(max_n, nseq, 1), dim3(256), 0, st, b);
}

// the generic 2-loop sums of the listed cells of `count` (<= 3) diagonals from g0 on (inside: upwards,
// outside: downwards): see k_tlane_gen for when a batch may run
void launch_tlane_gen(const TreeBatch& b, bool contra, bool outside, uint32_t g0, uint32_t count, uint32_t max_n,
                      uint32_t nseq, hipStream_t st) {
  if (count == 0u || nseq == 0u) return;
  const uint32_t dmin = outside ? g0 - (count - 1u) : g0;  // the longest diagonal of the batch
  if (dmin >= max_n) return;
  const dim3 grid(((max_n - dmin + 63u) / 64u + 1u) / 2u, nseq, 1), block(64 * kLaneParts * kGenDiags);
  if (contra) {
    if (outside) hipLaunchKernelGGL((k_tlane_gen<true, true>), grid, block, 0, st, b, g0, count);
    else hipLaunchKernelGGL((k_tlane_gen<true, false>), grid, block, 0, st, b, g0, count);
  } else {
    if (outside) hipLaunchKernelGGL((k_tlane_gen<false, true>), grid, block, 0, st, b, g0, count);
    else hipLaunchKernelGGL((k_tlane_gen<false, false>), grid, block, 0, st, b, g0, count);
  }
}

// d: the diagonal of the per-row role (>= max_n: none); d_b: the diagonal whose listed cells take the
// just-in-time part of their 2-loop sums in this launch (>= max_n: none)
void launch_tlane_outside(const TreeBatch& b, bool contra, uint32_t d, uint32_t d_b, uint32_t max_n, uint32_t nseq,
                          uint32_t thr, hipStream_t st) {
  const uint32_t nb_a = d < max_n ? (max_n - d + 64u * kLaneParts - 1u) / (64u * kLaneParts) : 0u;
  const uint32_t nb_b = d_b < max_n ? (max_n - d_b + 64u * kLaneParts - 1u) / (64u * kLaneParts) : 0u;
  if (nb_a + nb_b == 0u || nseq == 0u) return;
  if (contra)
    hipLaunchKernelGGL(k_tlane_outside<true>, dim3(nb_a + nb_b, nseq, 1), dim3(64 * kLaneParts), 0, st, b, d, thr, nb_a, d_b);
  else
    hipLaunchKernelGGL(k_tlane_outside<false>, dim3(nb_a + nb_b, nseq, 1), dim3(64 * kLaneParts), 0, st, b, d, thr, nb_a, d_b);
}

void launch_tlane_inside(const TreeBatch& b, bool contra, uint32_t d, uint32_t d_b, uint32_t max_n, uint32_t nseq,
                         uint32_t thr, hipStream_t st) {
  const uint32_t nb_a = d < max_n ? (max_n - d + 64u * kLaneParts - 1u) / (64u * kLaneParts) : 0u;
  const uint32_t nb_b = d_b < max_n ? (max_n - d_b + 64u * kLaneParts - 1u) / (64u * kLaneParts) : 0u;
  if (nb_a + nb_b == 0u || nseq == 0u) return;
  if (contra)
    hipLaunchKernelGGL(k_tlane_inside<true>, dim3(nb_a + nb_b, nseq, 1), dim3(64 * kLaneParts), 0, st, b, d, thr, nb_a, d_b);
  else
    hipLaunchKernelGGL(k_tlane_inside<false>, dim3(nb_a + nb_b, nseq, 1), dim3(64 * kLaneParts), 0, st, b, d, thr, nb_a, d_b);
}
