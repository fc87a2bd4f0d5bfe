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

// (+) of A[oa(x)] + B[ob(x)] over this wave's steps of x in [lo, hi), x < mine (the lane's own end)
template <typename FA, typename FB>
__device__ __forceinline__ void lane_sum(Acc& acc, const float* __restrict__ A, const float* __restrict__ B, uint32_t lo,
                                         uint32_t hi, uint32_t mine, uint32_t part, FA oa, FB ob) {
  for (uint32_t x = lo + 8u * part; x < hi; x += 8u * kLaneParts) {
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
    acc_add4(acc, v[0], v[1], v[2], v[3]);
    acc_add4(acc, v[4], v[5], v[6], v[7]);
  }
}

// The generic 2-loops of one cell, class by class: this wave's steps of the slots e < cnt of the
// class's list (ordered by a + b), the class's plane at diagonal `dbase -/+ (a + b)`, row
// `i +/- (1 + a)`; st4: the cell's own class scores; own: added to every term (outside: sums_close of
// the cell; there a slot counts for the lanes whose row has room for it, a < i and b < room = n - 1 - j).
// A step's scalar work is what a wave spends most issue slots on here (the first form of this loop
// took ~25 scalar instructions a slot for 64-bit plane and row offsets): the plane is the loop's
// constant, the slot's row a 32-bit float offset inside it, the lane's part of the address one
// register for the whole kernel.
template <bool OUTSIDE, bool TAIL>
__device__ __forceinline__ void lane_generic_step(Acc& acc, const float* __restrict__ plane, const uint32_t* __restrict__ gs,
                                                  const float* __restrict__ gl, uint32_t e, uint32_t cnt, uint32_t ld,
                                                  uint32_t dbase, uint32_t i, uint32_t vidx, float add, uint32_t room) {
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
    if (OUTSIDE) {
      okl[u] = a < i && sab - a < room;
      g[u] = (plane + ((dbase + sab) * ld - a))[vidx];  // (vidx = max(i, 1) - 1: a lane without room reads a neighbour)
    } else {
      g[u] = (plane + ((dbase - sab) * ld + a))[vidx];  // (vidx = i + 1)
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  float x[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const float v = (g[u] + ln8[u]) + add;
    x[u] = ((!TAIL || e + static_cast<uint32_t>(u) < cnt) && okl[u]) ? v : kNegInf;
  }
  acc_add4(acc, x[0], x[1], x[2], x[3]);
  acc_add4(acc, x[4], x[5], x[6], x[7]);
}
template <bool CONTRA, bool OUTSIDE>
__device__ __forceinline__ void lane_generic(Acc& acc, const TreeBatch& b, const float* __restrict__ x4, size_t msz,
                                             uint32_t ld, uint32_t smax, uint32_t dbase, uint32_t i, const float4& st4,
                                             float own, uint32_t room, uint32_t part) {
  const uint32_t vidx = OUTSIDE ? max(i, 1u) - 1u : i + 1u;
  uint32_t turn = part;  // (the classes' steps are dealt to the block's waves in one round-robin)
#pragma unroll
  for (uint32_t c = 0; c < 4u; c++) {
    const uint32_t start = sload(&b.tabs->gstart[CONTRA ? 1 : 0][c]);
    const uint32_t cnt = sload(&b.tabs->gcount[CONTRA ? 1 : 0][c][smax]);
    const uint32_t* __restrict__ gs = b.tabs->gslot[CONTRA ? 1 : 0] + start;
    const float* __restrict__ gl = b.tabs->glen[CONTRA ? 1 : 0] + start;
    const float* __restrict__ plane = x4 + c * msz;
    const float add = OUTSIDE ? own + pick(st4, c) : pick(st4, c);
    const uint32_t steps = (cnt + 7u) >> 3, full = cnt >> 3;
    uint32_t st = turn;
    for (; st < full; st += kLaneParts)
      lane_generic_step<OUTSIDE, false>(acc, plane, gs, gl, 8u * st, cnt, ld, dbase, i, vidx, add, room);
    if (st < steps) lane_generic_step<OUTSIDE, true>(acc, plane, gs, gl, 8u * st, cnt, ld, dbase, i, vidx, add, room);
    turn = (turn + kLaneParts - steps % kLaneParts) % kLaneParts;
  }
}

// the block's partial sums of NA accumulators meet in its first wave (false: this wave is done)
template <int NA>
__device__ __forceinline__ bool lane_join(Acc (&a)[NA], float2 (*lds)[kLaneParts][64], uint32_t lane, uint32_t part) {
  if (kLaneParts == 1u) return true;
  if (part != 0u) {
#pragma unroll
    for (int x = 0; x < NA; x++) lds[x][part][lane] = make_float2(a[x].m, a[x].s);
  }
  __syncthreads();
  if (part != 0u) return false;
#pragma unroll
  for (int x = 0; x < NA; x++)
    for (uint32_t p2 = 1; p2 < kLaneParts; p2++) {
      const float2 v = lds[x][p2][lane];
      acc_merge(a[x], Acc{v.x, v.y});
    }
  return true;
}

// ---- inside sweep, diagonal d (src/mccaskill_algo.rs:296-351 / 430-486)
template <bool CONTRA>
__global__ void __launch_bounds__(64 * kLaneParts) k_tlane_inside(TreeBatch b, uint32_t d, uint32_t thr) {
  __shared__ float2 red[2][kLaneParts][64];
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t part = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)));
  const uint32_t i = blockIdx.x * 64u + lane;
  if (blockIdx.x * 64u + d >= n) return;  // (the whole block: no barrier is left behind)
  const bool valid = i + d < n;
  const uint32_t j = i + d;
  const size_t msz = static_cast<size_t>(q.m[1] - q.m[0]);
  const size_t row = static_cast<size_t>(i) * ld + j, col = static_cast<size_t>(j) * ld + i;
  const size_t dg = static_cast<size_t>(d) * ld + i;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // [0] closing-pair block, [1] sums_multibranch
  Acc acc[2] = {acc_empty(), acc_empty()};
  const float mbc = valid ? q.m[T_MBC][row] : kNegInf;
  if (mbc > kNegInf) {
    if (part == 0u) {
      const float hp = q.m[T_HP][row];
      const float4 n4 = reinterpret_cast<const float4*>(q.m[T_NEAR4])[row];
      const float4 n8 = CONTRA ? zero4 : reinterpret_cast<const float4*>(q.m[T_NEAR8])[row];
      const float qm = d >= 2u ? q.m[T_QM][row + ld - 1u] : kNegInf;  // Qm(i+1, j-1)
      const float nr[8] = {n4.x, n4.y, n4.z, n4.w, n8.x, n8.y, n8.z, 0.f};
      float xs[8];
#pragma unroll
      for (uint32_t t = 0; t < 8u; t++) {
        xs[t] = kNegInf;
        if (t < Special<CONTRA>::N) {
          uint32_t a, bb;
          Special<CONTRA>::slot(t, a, bb);
          if (a + bb + 3u <= d) xs[t] = q.m[T_QB_D][static_cast<size_t>(d - 2u - a - bb) * ld + (i + 1u + a)] + nr[t];
        }
      }
      acc_add4(acc[0], hp, qm + mbc, xs[0], xs[1]);
      if (CONTRA) {
        acc_add2(acc[0], xs[2], xs[3]);
      } else {
        acc_add4(acc[0], xs[2], xs[3], xs[4], xs[5]);
        acc_add(acc[0], xs[6]);
      }
    }
    if (d >= 5u) {  // (a generic slot has a + b >= 2)
      const float4 cs = reinterpret_cast<const float4*>(q.m[T_CS4])[row];
      lane_generic<CONTRA, false>(acc[0], b, q.m[T_X4], msz, ld, min(d - 3u, 30u), d - 2u, i, cs, 0.f, 0u, part);
    }
  }
  // sums_multibranch: x = Q1's span, Q1(i, i+x) + Zr_mb(i+1+x, j); banded (thr != 0): the terms with
  // both spans below thr are k_tree_mid's
  if (d >= 2u) {
    const float* __restrict__ A = q.m[T_Q1_D] + i;
    const float* __restrict__ B = q.m[T_ZRM_D] + (i + 1u);
    auto oa = [&](uint32_t x) { return static_cast<size_t>(x) * ld; };
    auto ob = [&](uint32_t x) { return static_cast<size_t>(d - 1u - x) * ld + x; };
    const uint32_t mine = valid ? n : 0u;
    if (thr != 0u) {
      lane_sum(acc[1], A, B, 0u, d - thr, mine, part, oa, ob);
      lane_sum(acc[1], A, B, thr, d - 1u, mine, part, oa, ob);
    } else {
      lane_sum(acc[1], A, B, 0u, d - 1u, mine, part, oa, ob);
    }
  }
  if (!lane_join<2>(acc, red, lane, part)) return;
  if (!valid) return;
  if (thr != 0u && d >= 2u) {
    const float2 mm = q.mid[static_cast<size_t>(d % b.ring) * q.vec + i];
    acc_merge(acc[1], Acc{mm.x, mm.y});
  }
  float qa = kNegInf;
  if (mbc > kNegInf) {
    const float qb = acc_value(acc[0]);
    if (qb > kNegInf) {
      qa = qb + q.m[T_ACCS][row];
      const float4 in4 = reinterpret_cast<const float4*>(q.m[T_IN4])[row];
      q.m[T_QB][row] = qb;
      q.m[T_QB_D][dg] = qb;
      q.m[T_QA][row] = qa;
      float* __restrict__ x4 = q.m[T_X4];
      x4[dg] = qb + in4.x;
      x4[msz + dg] = qb + in4.y;
      x4[2u * msz + dg] = qb + in4.z;
      x4[3u * msz + dg] = qb + in4.w;
    }
  }
  const float ext_bp = CONTRA ? b.params->contra.external_score_basepair : 0.f;
  const float ext_un = CONTRA ? b.params->contra.external_score_unpair : 0.f;
  const float mb_bp = CONTRA ? b.params->contra.multibranch_score_basepair : b.params->turner.coeff_num_branches;
  const float mb_un = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  // rightmost-pair sums along the row, their column prefix
  const float zr_e_prev = j >= 1u ? q.m[T_ZRE][col - ld] : kNegInf;              // Zr_ext(i, j-1)
  const float zr_m_prev = (CONTRA && j >= 1u) ? q.m[T_ZRM][col - ld] : kNegInf;  // Zr_mb(i, j-1)
  const float u_next = q.m[T_U][col + 1u];                                        // U(i+1, j) (i+1 == n: the pad)
  const float zr_e = lse2(zr_e_prev + ext_un, qa + ext_bp);
  const float zr_m = CONTRA ? lse2(zr_m_prev + mb_un, qa + mb_bp) : zr_e + mb_bp;
  const float u = lse2(u_next + mb_un, zr_m);
  const float qmv = acc_value(acc[1]);
  const float q1 = lse2(u, qmv);
  q.m[T_ZRE][col] = zr_e;
  q.m[T_ZRM][col] = zr_m;
  q.m[T_ZRM_D][dg] = zr_m;
  q.m[T_U][col] = u;
  q.m[T_QM][row] = qmv;
  q.m[T_Q1R][row] = q1;
  q.m[T_Q1C][col] = q1;
  q.m[T_Q1_D][dg] = q1;
}

// ---- outside sweep, diagonal d, from the top (src/mccaskill_algo.rs:528-606 / 640-720).  In this sweep
// T_QM holds probs_multibranch2 and T_U the column prefix of probs_multibranch DIAGONAL-major (their
// only readers are this kernel's neighbours), T_X4 the planes PX4 = (log bpp - sums_close) + CS4[class],
// T_ZRM_D R = Pm (+) Pm2, T_W_D W = (log bpp + mbclose) - sums_close; W row-major (T_ZRE) and R
// column-major (T_ZRM) are kept for k_tree_mid.
template <bool CONTRA>
__global__ void __launch_bounds__(64 * kLaneParts) k_tlane_outside(TreeBatch b, uint32_t d, uint32_t thr) {
  __shared__ float2 red[3][kLaneParts][64];
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t part = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)));
  const uint32_t i = blockIdx.x * 64u + lane;
  if (blockIdx.x * 64u + d >= n) return;
  const bool valid = i + d < n;
  const uint32_t j = i + d;
  const size_t msz = static_cast<size_t>(q.m[1] - q.m[0]);
  const size_t row = static_cast<size_t>(i) * ld + j, col = static_cast<size_t>(j) * ld + i;
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

  // [0] probs_multibranch, [1] the pair's 2-loop terms, [2] L_e cases one and three
  Acc acc[3] = {acc_empty(), acc_empty(), acc_empty()};
  // probs_multibranch(i,j) (540-543): x = 1 .., W(i, j+1+x) + Q1(j+1, j+x); banded: W's span d+1+x < thr
  const uint32_t hi = thr != 0u ? (thr > d + 1u ? thr - 1u - d : 0u) : n - d;  // (uniform end; room, i < n - d)
  if (hi > 1u) {
    auto oa = [&](uint32_t x) { return static_cast<size_t>(d + 1u + x) * ld; };
    auto ob = [&](uint32_t x) { return static_cast<size_t>(x - 1u) * ld + d + 1u; };
    lane_sum(acc[0], w_d + i, q1_d + i, 1u, hi, room, part, oa, ob);
  }
  if (paired) {
    // enclosing 2-loops (562-593)
    if (part == 0u) {
      float xs[8];
#pragma unroll
      for (uint32_t t = 0; t < 8u; t++) {
        xs[t] = kNegInf;
        if (t < Special<CONTRA>::N) {
          uint32_t a, bb;
          Special<CONTRA>::slot(t, a, bb);
          if (a < i && bb < room) {
            const uint32_t k = i - 1u - a, dd = d + 2u + a + bb;
            const float nqb = q.m[T_QB_D][static_cast<size_t>(dd) * ld + k];
            const float npk = q.out[tri_off(n, dd) + k];
            const float nsc = (t < 4u ? q.m[T_NEAR4] : q.m[T_NEAR8])[4u * (static_cast<size_t>(k) * ld + (k + dd)) + (t & 3u)];
            if (nqb > kNegInf) xs[t] = ((npk + qb) - nqb) + nsc;
          }
        }
      }
      acc_add4(acc[1], xs[0], xs[1], xs[2], xs[3]);
      if (!CONTRA) acc_add4(acc[1], xs[4], xs[5], xs[6], kNegInf);
    }
    if (n >= d + 5u) {  // (a generic slot has a + b >= 2, and a + b <= (i - 1) + room = n - 3 - d)
      const float4 in4 = reinterpret_cast<const float4*>(q.m[T_IN4])[row];
      lane_generic<CONTRA, true>(acc[1], b, q.m[T_X4], msz, ld, min(n - 3u - d, 30u), d + 2u, i, in4, qb, room, part);
    }
  }
  // L_e cases one and three (594-601): x = 1 .., Q1(i-x, i-1) + R(i-1-x, j); banded: R's span d+1+x < thr
  if (hi > 1u) {
    auto oa = [&](uint32_t x) { return static_cast<size_t>(x - 1u) * ld - x; };
    auto ob = [&](uint32_t x) { return static_cast<size_t>(d + 1u + x) * ld - 1u - x; };
    lane_sum(acc[2], q1_d + i, r_d + i, 1u, hi, paired ? i : 0u, part, oa, ob);
  }
  if (!lane_join<3>(acc, red, lane, part)) return;
  if (!valid) return;
  if (thr != 0u) {
    const float2 m1 = q.mid[(static_cast<size_t>(b.ring) + d % b.ring) * q.vec + i];
    acc_merge(acc[0], Acc{m1.x, m1.y});
    const float2 m2 = q.mid[(2u * static_cast<size_t>(b.ring) + d % b.ring) * q.vec + i];
    acc_merge(acc[2], Acc{m2.x, m2.y});
  }
  // probs_multibranch2(i,j) from the right neighbour (544-549), R, the column prefix
  const float pm2_next = room >= 1u ? pm2_d[dg1] : kNegInf, w_next = room >= 1u ? w_d[dg1] : kNegInf;
  const float pm2 = lse2(pm2_next + mb_un, w_next);
  const float pm = acc_value(acc[0]);
  const float r = lse2(pm, pm2);
  const float sp_prev = i >= 1u ? sp_d[dg1 - 1u] : kNegInf;  // prefix of column j up to row i-1: cell (i-1, j)
  pm2_d[dg] = pm2;
  r_d[dg] = r;
  q.m[T_ZRM][col] = r;
  sp_d[dg] = lse2(sp_prev + mb_un, pm);
  if (!paired) return;
  // the pair's probability (562-604): external, enclosing 2-loops, multibranch cases
  const float qa = q.m[T_QA][row];
  const float zpi = q.zp[i], zsj = q.zs[j + 1u], ztot = sload(q.zp + n);
  acc_add(acc[1], CONTRA ? (((zpi + zsj) + qa) + ext_bp) - ztot : ((zpi + qa) + zsj) - ztot);
  const float A = qa + abr;
  acc_add(acc[1], A + acc_value(acc[2]));
  acc_add(acc[1], A + sp_prev);
  const float lp = acc_value(acc[1]);
  if (lp > kNegInf) {
    const float w = (lp + q.m[T_MBC][row]) - qb;
    const float4 cs = reinterpret_cast<const float4*>(q.m[T_CS4])[row];
    const float pq = lp - qb;
    q.out[tri_off(n, d) + i] = lp;
    q.m[T_ZRE][row] = w;
    w_d[dg] = w;
    float* __restrict__ x4 = q.m[T_X4];
    x4[dg] = pq + cs.x;
    x4[msz + dg] = pq + cs.y;
    x4[2u * msz + dg] = pq + cs.z;
    x4[3u * msz + dg] = pq + cs.w;
  }
}

}  // namespace

void launch_tlane_outside(const TreeBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq, uint32_t thr,
                          hipStream_t st) {
  const uint32_t gx = (max_n - d + 63u) / 64u;
  if (contra)
    hipLaunchKernelGGL(k_tlane_outside<true>, dim3(gx, nseq, 1), dim3(64 * kLaneParts), 0, st, b, d, thr);
  else
    hipLaunchKernelGGL(k_tlane_outside<false>, dim3(gx, nseq, 1), dim3(64 * kLaneParts), 0, st, b, d, thr);
}

void launch_tlane_inside(const TreeBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq, uint32_t thr,
                         hipStream_t st) {
  const uint32_t gx = (max_n - d + 63u) / 64u;
  if (contra)
    hipLaunchKernelGGL(k_tlane_inside<true>, dim3(gx, nseq, 1), dim3(64 * kLaneParts), 0, st, b, d, thr);
  else
    hipLaunchKernelGGL(k_tlane_inside<false>, dim3(gx, nseq, 1), dim3(64 * kLaneParts), 0, st, b, d, thr);
}
