// rnamc_tree_band.h — tree-order mode: BANDS OF ANTI-DIAGONALS PER LAUNCH (included by
// rnamc_tree.hip inside its anonymous namespace; it uses that file's Acc / DPP / TSeq helpers).
//
// Why.  The banded sweep of rnamc_tree.hip is a chain of dependent launches, two anti-diagonals
// each, ~11 us per launch at n = 4096 whatever the launch computes (dispatch, operand round
// trips, reduction, epilogue, drain): 4 092 launches = the 49.5 ms of round 3, with HBM at 0.26 and
// the VALUs at 0.11 of their peaks.  Here ONE launch sweeps kBB = 8 anti-diagonals.
//
// How.  A workgroup of 1 024 threads owns kBR = 25 rows of the band and computes a TRAPEZOID: at
// step s (diagonal d0 + s inside, d0 - s outside) the cells of its own rows plus kBB-1-s ghost rows
// of its neighbour (inside: the rows after its own; outside: the rows before), because every cell
// (i, d) of the recurrences (src/mccaskill_algo.rs:282-723) depends only on cells (i + a, d - a - b)
// (inside) resp. (i - a, d + a + b) (outside), a, b >= 0.  So a workgroup never reads what another
// workgroup computes in the same launch — no grid barrier, no flag, no inter-workgroup visibility
// question; the ghost cells are computed twice (their owner stores them), 28 % more cells.
//   phase A (bulk, off the chain): for every (step, cell) of the trapezoid the part of its sums
//     whose operands were final BEFORE the launch — the <= 496 2-loop probes whose inner / outer
//     pair lies outside the band, the edge terms of the cubic products (<= 4 mid-field band widths,
//     rnamc_tree.hip k_tree_mid) between operands of older launches, the mid-field ring entry — as
//     {max, sum} pairs in LDS.  A half-wave (32 lanes) per cell, all steps of a cell in one
//     half-wave; probe slots are walked in CLASS order (TreeTabs::slot_*), so a gather's plane and
//     the component of the cell's static it adds are uniform over the wave; the special small
//     loops take their scores from per-cell statics (T_NEAR4 / T_NEAR8), no scorer runs here.
//   phase B (the chain): kBB steps separated by a workgroup barrier; a step adds the few terms
//     with an operand computed IN the band (read from LDS: <= 21 probes, <= 15 product terms, the
//     neighbours' scalars), reduces over the half-wave, runs the scalar recurrences and stores.
// A step of the chain touches no global memory before its stores.
//
// Used for the banded part of a sweep (thr != 0: diagonals >= 3 band widths inside, below
// n - 2 band widths outside); the first / last diagonals keep the two-diagonal launches.

constexpr int kBB = 8;                  // anti-diagonals per launch
constexpr int kBC = 32;                 // cell slots of a workgroup (one half-wave each)
constexpr int kBR = kBC - (kBB - 1);    // rows a workgroup owns

struct BandSlot {
  int rel;      // float offset of the probed pair from the cell, inside a plane
  float len;    // length-dependent score of the slot
  int sab;      // a + b, or 1 << 20 for a padding slot
  int pad;
};
struct BandShared {
  BandSlot slot[kSlotGroups * 32];
  float ib[kBB + 2][kBC + 2][8];   // cells computed in the band (+ halo steps -2, -1; rows -1 .. kBC)
  float2 pre[kBB][kBC][3];         // phase A: {max, sum} of the sums' out-of-band parts
  float st[kBB][kBC][20];          // per-cell statics
  float part[kBB][kBC][16];        // out-of-band partners of the in-band product terms
  float lenab[8][8];
};
// st[] layout
constexpr int ST_MBC = 0, ST_ACCS = 1, ST_HP = 2, ST_EXT = 3, ST_CS4 = 4, ST_IN4 = 8, ST_NEAR = 12,
              ST_QB = 19;  // (outside: [1] = sums_accessible, [2] unused, [3] exterior term, [19] sums_close)
// ib[] fields
constexpr int IN_QB = 0, IN_QM = 1, IN_ZRE = 2, IN_ZRM = 3, IN_U = 4, IN_Q1 = 5;
constexpr int OUT_LP = 0, OUT_W = 1, OUT_PM2 = 2, OUT_R = 3, OUT_SP = 4, OUT_PQ = 5;

// reductions over the 32 lanes of a half-wave (lanes 0..31 and 32..63 separately); the result is
// returned in every lane of the half
__device__ __forceinline__ float half_pick(float v) {
  const float lo = __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(v)), 31)));
  const float hi = __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(v)), 63)));
  return (threadIdx.x & 32u) ? hi : lo;
}
__device__ __forceinline__ float half_max(float v) {
  v = vmaxf(v, dpp<0x111, 0xF>(kNegInf, v));
  v = vmaxf(v, dpp<0x112, 0xF>(kNegInf, v));
  v = vmaxf(v, dpp<0x114, 0xF>(kNegInf, v));
  v = vmaxf(v, dpp<0x118, 0xF>(kNegInf, v));
  v = vmaxf(v, dpp<0x142, 0xA>(kNegInf, v));
  return half_pick(v);
}
__device__ __forceinline__ float half_sum(float v) {
  v += dpp<0x111, 0xF>(0.f, v);
  v += dpp<0x112, 0xF>(0.f, v);
  v += dpp<0x114, 0xF>(0.f, v);
  v += dpp<0x118, 0xF>(0.f, v);
  v += dpp<0x142, 0xA>(0.f, v);
  return half_pick(v);
}
__device__ __forceinline__ Acc half_reduce(const Acc& a) {
  const float m = half_max(a.m);
  const float s = half_sum(a.s * ex2((a.m - m) * kL2E));
  return Acc{m, s};
}
__device__ __forceinline__ float2 acc_pack(const Acc& a) { return make_float2(a.m, a.s); }
__device__ __forceinline__ Acc acc_unpack(const float2& v) { return Acc{v.x, v.y}; }

// index of the special slot (a, b) in Special<CONTRA>::slot's order
template <bool CONTRA>
__device__ __forceinline__ uint32_t special_index(uint32_t a, uint32_t bb) {
  if (CONTRA) return 2u * a + bb;
  return a == 0u ? bb : (a == 1u ? 2u + bb : 4u + bb);  // (0,0) (0,1) | (1,0) (1,1) (1,2) | (2,1) (2,2)
}

// slot tables of the launch's model into LDS: per-lane constants of the class-ordered probe list
template <bool OUTSIDE>
__device__ __forceinline__ void band_load_slots(BandShared& S, const TreeBatch& b, int model, uint32_t ld) {
  for (uint32_t x = threadIdx.x; x < kSlotGroups * 32u; x += blockDim.x) {
    const uint32_t ab = b.tabs->slot_ab[model][x];
    const int a = static_cast<int>(ab & 255u), bb = static_cast<int>((ab >> 8) & 255u);
    BandSlot sl;
    const int r = (1 + a) * static_cast<int>(ld) - (1 + bb);
    sl.rel = OUTSIDE ? -r : r;
    sl.len = b.tabs->slot_len[model][x];
    sl.sab = (ab >> 16) ? a + bb : (1 << 20);
    sl.pad = (a << 8) | bb;
    S.slot[x] = sl;
  }
  if (threadIdx.x < 64u) S.lenab[threadIdx.x >> 3][threadIdx.x & 7u] = b.tabs->len_ab[model][threadIdx.x >> 3][threadIdx.x & 7u];
}

#define IB(step, row) S.ib[static_cast<int>(step) + 2][static_cast<int>(row) + 1]

// two-stage software pipeline over the steps of a band: the loads of step s + 1 (issue) are in
// flight while step s is summed (consume); issue is called for one step past the last (it then
// reads in-bounds filler)
template <class Loads, class Issue, class Consume>
__device__ __forceinline__ void band_pipeline(uint32_t nst, Issue&& issue, Consume&& consume) {
  Loads La, Lb;
  issue(La, 0u);
  for (uint32_t s = 0; s < nst; s += 2u) {
    issue(Lb, s + 1u);
    consume(La, s);
    if (s + 1u >= nst) break;
    issue(La, s + 2u);
    consume(Lb, s + 1u);
  }
}

// ------------------------------------------------------------------------------------------
// inside sweep, diagonals d0 .. d0 + nsteps - 1 (all in one mid-field band: thr is theirs)
template <bool CONTRA>
__global__ void __launch_bounds__(1024) k_tree_band_in(TreeBatch b, uint32_t d0, uint32_t nsteps, uint32_t thr) {
  __shared__ BandShared S;
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  if (d0 >= n) return;
  const uint32_t r0 = blockIdx.x * kBR;
  if (r0 + d0 >= n) return;  // (uniform: no cell of this tile exists)
  const uint32_t nst = min(nsteps, n - d0);
  const uint32_t c = threadIdx.x >> 5, l = threadIdx.x & 31u;
  const uint32_t i = r0 + c;
  const size_t row_i = static_cast<size_t>(i) * ld;
  const size_t msz = static_cast<size_t>(q.m[1] - q.m[0]);
  const float ext_bp = CONTRA ? b.params->contra.external_score_basepair : 0.f;
  const float ext_un = CONTRA ? b.params->contra.external_score_unpair : 0.f;
  const float mb_bp = CONTRA ? b.params->contra.multibranch_score_basepair : b.params->turner.coeff_num_branches;
  const float mb_un = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;

  band_load_slots<false>(S, b, CONTRA ? 1 : 0, ld);
  // (cells that do not exist — past a row's end, outside the trapezoid — read as "absent")
  for (uint32_t x = threadIdx.x; x < (kBB + 2) * (kBC + 2) * 8; x += blockDim.x) (&S.ib[0][0][0])[x] = kNegInf;
  __syncthreads();
  // halo: what steps 0 and 1 read of the two diagonals below the band (rows 0 .. kBC)
  if (threadIdx.x < 2u * (kBC + 2)) {
    const uint32_t hs = threadIdx.x / (kBC + 2), rl = threadIdx.x % (kBC + 2);  // row rl - 1
    const uint32_t dd = d0 - 2u + hs;
    const uint32_t ii = r0 + rl - 1u, jj = ii + dd;
    const bool ok = rl >= 1u && jj < n;
    float* cell = S.ib[hs][rl];
    cell[IN_QB] = kNegInf;
    cell[IN_Q1] = kNegInf;
    cell[IN_QM] = ok ? q.m[T_QM][static_cast<size_t>(ii) * ld + jj] : kNegInf;
    cell[IN_ZRE] = ok ? q.m[T_ZRE][static_cast<size_t>(jj) * ld + ii] : kNegInf;
    cell[IN_ZRM] = ok ? q.m[T_ZRM][static_cast<size_t>(jj) * ld + ii] : kNegInf;
    cell[IN_U] = ok ? q.m[T_U][static_cast<size_t>(jj) * ld + ii] : kNegInf;
  }
  __syncthreads();

  // ---- phase A.  Everything here reads data of EARLIER launches only, so nothing orders the steps:
  // the statics of all steps first (one round trip), then a two-stage software pipeline — the
  // loads of step s+1 are in flight while step s is summed (a half-wave that waited for its own
  // loads step by step spent two round trips per step: 69 us per launch instead of ~20).
  const float* __restrict__ A1 = q.m[T_Q1R] + row_i + i;  // A1[s1] = Q1(i, i + s1)
  const float* __restrict__ qx = q.m[T_X4];
  for (uint32_t s = 0; s < nst; s++) {
    const uint32_t j = i + d0 + s;
    const bool valid = c + s < kBC && j < n;
    const size_t o = row_i + j;
    if (l < 20u) {
      float v = 0.f;
      if (valid) {
        if (l == ST_MBC) v = q.m[T_MBC][o];
        else if (l == ST_ACCS) v = q.m[T_ACCS][o];
        else if (l == ST_HP) v = q.m[T_HP][o];
        else if (l >= ST_CS4 && l < ST_CS4 + 4) v = q.m[T_CS4][4u * o + (l - ST_CS4)];
        else if (l >= ST_IN4 && l < ST_IN4 + 4) v = q.m[T_IN4][4u * o + (l - ST_IN4)];
        else if (l >= ST_NEAR && l < ST_NEAR + 4) v = q.m[T_NEAR4][4u * o + (l - ST_NEAR)];
        else if (l >= ST_NEAR + 4 && l < ST_NEAR + 7) v = q.m[T_NEAR8][4u * o + (l - ST_NEAR - 4)];
      } else if (l == ST_MBC) {
        v = kNegInf;
      }
      S.st[s][c][l] = v;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // two pipelines, one after the other (one pipeline holding both kinds of loads of two steps
  // needs more than the 128 registers a 1024-thread workgroup has): the 2-loop probes, then the
  // edge terms of sums_multibranch
  struct InProbes {
    float g[kSlotGroups];
    float sp;
  };
  auto issue_p = [&](InProbes& L, uint32_t s) {
    const uint32_t d = d0 + s, j = i + d;
    const bool valid = s < nst && c + s < kBC && j < n;
    const size_t o = valid ? row_i + j : 0;  // (no cell: every load below reads in-bounds filler)
    const bool act = valid && S.st[s < nst ? s : 0][c][ST_MBC] > kNegInf;
    const int sm2 = static_cast<int>(s) - 2;
#pragma unroll
    for (int g = 0; g < kSlotGroups; g++) {
      const BandSlot sl = S.slot[g * 32 + static_cast<int>(l)];
      const bool ok = act && sl.sab + 3 <= static_cast<int>(d) && sl.sab > sm2;
      const uint32_t cls = slot_group_class(static_cast<uint32_t>(g));
      L.g[g] = qx[cls * msz + (ok ? o + static_cast<size_t>(static_cast<int64_t>(sl.rel)) : o)];
    }
    uint32_t a = 0, bb = 0;
    if (l < Special<CONTRA>::N) Special<CONTRA>::slot(l, a, bb);
    const bool ok = act && l < Special<CONTRA>::N && a + bb + 3u <= d && static_cast<int>(a + bb) > sm2;
    L.sp = q.m[T_QB][ok ? static_cast<size_t>(i + 1u + a) * ld + (j - 1u - bb) : o];
  };
  auto consume_p = [&](const InProbes& L, uint32_t s) {
    const uint32_t d = d0 + s, j = i + d;
    const bool valid = c + s < kBC && j < n;
    const float* st = S.st[s][c];
    const bool act = valid && st[ST_MBC] > kNegInf;
    const int sm2 = static_cast<int>(s) - 2;
    Acc aq = acc_empty();
    if (act) {
      if (l == 0u) acc_add(aq, st[ST_HP]);
      if (l < Special<CONTRA>::N) {
        uint32_t a, bb;
        Special<CONTRA>::slot(l, a, bb);
        if (a + bb + 3u <= d && static_cast<int>(a + bb) > sm2) acc_add(aq, L.sp + st[ST_NEAR + l]);
      }
      float xv[kSlotGroups];
#pragma unroll
      for (int g = 0; g < kSlotGroups; g++) {
        const BandSlot sl = S.slot[g * 32 + static_cast<int>(l)];
        const bool ok = sl.sab + 3 <= static_cast<int>(d) && sl.sab > sm2;
        const uint32_t cls = slot_group_class(static_cast<uint32_t>(g));
        xv[g] = ok ? (L.g[g] + sl.len) + st[ST_CS4 + cls] : kNegInf;
      }
#pragma unroll
      for (int g = 0; g + 3 < kSlotGroups; g += 4) acc_add4(aq, xv[g], xv[g + 1], xv[g + 2], xv[g + 3]);
      acc_add(aq, xv[kSlotGroups - 1]);
    }
    aq = half_reduce(aq);
    if (l == 0u) S.pre[s][c][0] = acc_pack(aq);
  };
#ifdef RNAMC_DEBUG_KNOBS
  if (!(b.debug & 128))  // (timing experiments: results wrong)
#endif
  band_pipeline<InProbes>(nst, issue_p, consume_p);

  constexpr int kEdgeL = 5, kEdgeR = 4;  // chunks of 32 terms: left range <= 135 + 7, right <= 127
  struct InEdges {
    float ea[kEdgeL + kEdgeR], eb[kEdgeL + kEdgeR];
    float2 mid;
    float pt;
  };
  auto issue_e = [&](InEdges& L, uint32_t s) {
    const uint32_t d = d0 + s, j = i + d;
    const bool valid = s < nst && c + s < kBC && j < n;
    const float* __restrict__ B1 = q.m[T_ZRM] + (valid ? static_cast<size_t>(j) * ld + i + 1u : 0);  // B1[s1] = Zr_mb(i+1+s1, j)
    const float* __restrict__ A1v = valid ? A1 : q.m[T_Q1R];
    const uint32_t lhi = valid ? d - thr : 0u, rhi = valid ? min(d0, d - 1u) : 0u;
#pragma unroll
    for (int u = 0; u < kEdgeL; u++) {  // left: s1 in [s, d - thr)
      const uint32_t k = s + l + 32u * u;
      const uint32_t kk = k < lhi ? k : 0u;
      L.ea[u] = A1v[kk];
      L.eb[u] = B1[kk];
    }
#pragma unroll
    for (int u = 0; u < kEdgeR; u++) {  // right: s1 in [thr, min(d0, d - 1))
      const uint32_t k = thr + l + 32u * u;
      const uint32_t kk = k < rhi ? k : 0u;
      L.ea[kEdgeL + u] = A1v[kk];
      L.eb[kEdgeL + u] = B1[kk];
    }
    L.mid = q.mid[(static_cast<size_t>(0u) * b.ring + d % b.ring) * q.vec + (valid ? i : 0u)];
    // partners of the in-band terms (phase B): right t <= s-2: Zr_mb(i+1+d0+t, j); left t' <= s-1: Q1(i, i+s-1-t')
    const bool right = l < 8u && l + 2u <= s, left = l >= 8u && l < 16u && (l - 8u) + 1u <= s;
    const float* p = (valid && right) ? B1 + d0 + l : ((valid && left) ? A1 + (s - 1u - (l - 8u)) : q.m[T_QB]);
    L.pt = *p;
  };
  auto consume_e = [&](const InEdges& L, uint32_t s) {
    const uint32_t d = d0 + s, j = i + d;
    const bool valid = c + s < kBC && j < n;
    Acc am = acc_empty();
    if (valid) {
      if (l == 0u) am = acc_unpack(L.mid);
      const uint32_t lhi = d - thr, rhi = min(d0, d - 1u);
      float x[kEdgeL + kEdgeR];
#pragma unroll
      for (int u = 0; u < kEdgeL; u++) x[u] = (s + l + 32u * u < lhi) ? L.ea[u] + L.eb[u] : kNegInf;
#pragma unroll
      for (int u = 0; u < kEdgeR; u++) x[kEdgeL + u] = (thr + l + 32u * u < rhi) ? L.ea[kEdgeL + u] + L.eb[kEdgeL + u] : kNegInf;
      acc_add4(am, x[0], x[1], x[2], x[3]);
      acc_add4(am, x[4], x[5], x[6], x[7]);
      acc_add(am, x[8]);
      if ((l < 8u && l + 2u <= s) || (l >= 8u && l < 16u && (l - 8u) + 1u <= s)) S.part[s][c][l] = L.pt;
    }
    am = half_reduce(am);
    if (l == 0u) S.pre[s][c][1] = acc_pack(am);
  };
#ifdef RNAMC_DEBUG_KNOBS
  if (!(b.debug & 256))
#endif
  band_pipeline<InEdges>(nst, issue_e, consume_e);
  __syncthreads();

  // ---- phase B: the chain
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 512) return;
  if (b.debug & 1024) {  // the launch without any work
    return;
  }
#endif
  for (uint32_t s = 0; s < nst; s++) {
    const uint32_t d = d0 + s, j = i + d;
    const bool valid = c + s < kBC && j < n;
    if (valid) {
      const float* st = S.st[s][c];
      const float mbc = st[ST_MBC];
      const bool act = mbc > kNegInf;
      Acc aq = acc_empty(), am = acc_empty();
      if (act) {
        if (l == 0u) acc_add(aq, IB(s - 2, c + 1)[IN_QM] + mbc);  // Qm(i+1, j-1) + multibranch close
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
          const uint32_t tt = l + 32u * pass, a = tt >> 3, bb = tt & 7u;
          if (a + bb + 2u <= s) {
            const uint32_t cr = c + 1u + a, sp = s - 2u - a - bb;
            const float qbk = IB(sp, cr)[IN_QB];
            float x;
            if (Special<CONTRA>::is(a, bb)) {
              x = qbk + st[ST_NEAR + special_index<CONTRA>(a, bb)];
            } else {
              const uint32_t cls = slot_class(a, bb);
              x = ((qbk + S.st[sp][cr][ST_IN4 + cls]) + S.lenab[a][bb]) + st[ST_CS4 + cls];
            }
            acc_add(aq, x);
          }
        }
      }
      if (l < 8u) {
        if (l + 2u <= s) acc_add(am, IB(l, c)[IN_Q1] + S.part[s][c][l]);  // Q1(i, i+d0+l) + Zr_mb(.., j)
      } else if (l < 16u) {
        const uint32_t t = l - 8u;
        if (t + 1u <= s) acc_add(am, IB(t, c + s - t)[IN_ZRM] + S.part[s][c][l]);  // Zr_mb(k, j), span d0+t
      }
      aq = half_reduce(aq);
      am = half_reduce(am);
      acc_merge(aq, acc_unpack(S.pre[s][c][0]));
      acc_merge(am, acc_unpack(S.pre[s][c][1]));
      const float qb = act ? acc_value(aq) : kNegInf;
      const float qa = qb > kNegInf ? qb + st[ST_ACCS] : kNegInf;
      const float zr_e = lse2(IB(s - 1, c)[IN_ZRE] + ext_un, qa + ext_bp);
      const float zr_m = CONTRA ? lse2(IB(s - 1, c)[IN_ZRM] + mb_un, qa + mb_bp) : zr_e + mb_bp;
      const float u = lse2(IB(s - 1, c + 1)[IN_U] + mb_un, zr_m);
      const float qmv = acc_value(am);
      const float q1 = lse2(u, qmv);
      if (l == 0u) {
        float* cell = IB(s, c);
        cell[IN_QB] = qb;
        cell[IN_QM] = qmv;
        cell[IN_ZRE] = zr_e;
        cell[IN_ZRM] = zr_m;
        cell[IN_U] = u;
        cell[IN_Q1] = q1;
      }
      if (c < kBR) {  // the rows this workgroup owns: one store per lane
        const size_t o = row_i + j, oc = static_cast<size_t>(j) * ld + i;
        if (qb > kNegInf) {
          if (l == 0u) q.m[T_QB][o] = qb;
          else if (l == 1u) q.m[T_QA][o] = qa;
          else if (l >= 2u && l < 6u) q.m[T_X4][(l - 2u) * msz + o] = qb + st[ST_IN4 + (l - 2u)];
        }
        if (l == 6u) q.m[T_ZRE][oc] = zr_e;
        else if (l == 7u) q.m[T_ZRM][oc] = zr_m;
        else if (l == 8u) q.m[T_U][oc] = u;
        else if (l == 9u) q.m[T_QM][o] = qmv;
        else if (l == 10u) q.m[T_Q1R][o] = q1;
        else if (l == 11u) q.m[T_Q1C][oc] = q1;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// outside sweep, diagonals d0, d0 - 1, .., d0 - nsteps + 1 (all in one mid-field band)
template <bool CONTRA>
__global__ void __launch_bounds__(1024) k_tree_band_out(TreeBatch b, uint32_t d0, uint32_t nsteps, uint32_t thr) {
  __shared__ BandShared S;
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  const uint32_t dlast = d0 + 1u - nsteps;
  if (dlast >= n) return;
  const int r0 = static_cast<int>(blockIdx.x * kBR);
  if (r0 + static_cast<int>(dlast) >= static_cast<int>(n)) return;  // (uniform: no cell of this tile, at any step)
  const uint32_t c = threadIdx.x >> 5, l = threadIdx.x & 31u;
  const int ii = r0 - (kBB - 1) + static_cast<int>(c);  // row of this cell slot (may be negative: no cell)
  const uint32_t i = static_cast<uint32_t>(max(ii, 0));
  const size_t row_i = static_cast<size_t>(i) * ld;
  const size_t msz = static_cast<size_t>(q.m[1] - q.m[0]);
  const float mb_un = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  const float ext_bp = CONTRA ? b.params->contra.external_score_basepair : 0.f;
  const float abr = CONTRA ? b.params->contra.multibranch_score_basepair : b.params->turner.coeff_num_branches;
  const float* __restrict__ w_r = q.m[T_ZRE];  // W = (P + mbclose) - Qb, row-major
  float* __restrict__ r_c = q.m[T_ZRM];        // R = Pm (+) Pm2, column-major
  float* __restrict__ pm2_r = q.m[T_QM];       // probs_multibranch2, row-major
  float* __restrict__ sp_c = q.m[T_U];         // column prefix of Pm, column-major

  band_load_slots<true>(S, b, CONTRA ? 1 : 0, ld);
  for (uint32_t x = threadIdx.x; x < (kBB + 2) * (kBC + 2) * 8; x += blockDim.x) (&S.ib[0][0][0])[x] = kNegInf;
  __syncthreads();
  // halo: the diagonal above the band (rows -1 .. kBC - 1 of the slot numbering)
  if (threadIdx.x < kBC + 2) {
    const int rl = static_cast<int>(threadIdx.x) - 1;
    const int hi = r0 - (kBB - 1) + rl;
    const uint32_t hj = static_cast<uint32_t>(hi) + d0 + 1u;
    const bool ok = hi >= 0 && d0 + 1u < n && hj < n;
    float* cell = IB(-1, rl);
    const size_t ho = static_cast<size_t>(max(hi, 0)) * ld + hj;
    cell[OUT_LP] = kNegInf;
    cell[OUT_PQ] = kNegInf;
    cell[OUT_R] = kNegInf;
    cell[OUT_W] = ok ? w_r[ho] : kNegInf;
    cell[OUT_PM2] = ok ? pm2_r[ho] : kNegInf;
    cell[OUT_SP] = ok ? sp_c[static_cast<size_t>(hj) * ld + static_cast<uint32_t>(max(hi, 0))] : kNegInf;
  }
  __syncthreads();

  // ---- phase A (as in the inside kernel: statics of all steps, then a two-stage pipeline)
  const float* __restrict__ px = q.m[T_X4];
  const float zpi = (ii >= 0 && i <= n) ? q.zp[i] : 0.f, ztot = q.zp[n];
  for (uint32_t s = 0; s < nsteps; s++) {
    const uint32_t d = d0 - s, j = i + d;
    const bool valid = c >= s && ii >= 0 && j < n && d < n;
    const size_t o = row_i + j;
    if (l < 20u) {
      float v = 0.f;
      if (valid) {
        if (l == ST_MBC) v = q.m[T_MBC][o];
        else if (l == ST_ACCS) v = q.m[T_QA][o];
        else if (l == ST_HP) v = q.zs[j + 1u];  // (outside: Z(j+1, n-1) of the exterior term)
        else if (l == ST_QB) v = q.m[T_QB][o];
        else if (l >= ST_CS4 && l < ST_CS4 + 4) v = q.m[T_CS4][4u * o + (l - ST_CS4)];
        else if (l >= ST_IN4 && l < ST_IN4 + 4) v = q.m[T_IN4][4u * o + (l - ST_IN4)];
        else if (l >= ST_NEAR && l < ST_NEAR + 4) v = q.m[T_NEAR4][4u * o + (l - ST_NEAR)];
        else if (l >= ST_NEAR + 4 && l < ST_NEAR + 7) v = q.m[T_NEAR8][4u * o + (l - ST_NEAR - 4)];
      } else if (l == ST_QB) {
        v = kNegInf;
      }
      S.st[s][c][l] = v;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  constexpr int kEdge = 5;  // chunks of 32 terms: both ranges hold at most 2 * 64 + 7 terms
  auto geom = [&](uint32_t s, uint32_t& d, uint32_t& j, bool& valid, uint32_t& lim, uint32_t& base, uint32_t& hi) {
    d = d0 - s;
    j = i + d;
    valid = s < nsteps && c >= s && ii >= 0 && j < n && d < n;
    const uint32_t len0 = valid ? n - 1u - j : 0u;
    lim = thr != 0u ? min(len0, thr - 1u - d) : len0;               // probs_multibranch: x in [s, lim)
    base = (thr != 0u && j + 1u > thr) ? j + 1u - thr : 0u;         // L_e: k in [base, hi)
    hi = (valid && i > s) ? min(i - s, i - 1u) : 0u;                // (k = i-1 carries Q1(i, i-1) = -inf)
  };
  // three pipelines, one after the other (register budget): the enclosing 2-loops, the edge terms
  // of probs_multibranch, the edge terms of L_e
  struct OutProbes {
    float g[kSlotGroups];
    float sx, spk, ssc;
  };
  auto issue_p = [&](OutProbes& L, uint32_t s) {
    uint32_t d, j, lim, base, hi;
    bool valid;
    geom(s, d, j, valid, lim, base, hi);
    const size_t o = valid ? row_i + j : 0;
    const bool paired = valid && S.st[s < nsteps ? s : 0][c][ST_QB] > kNegInf;
    const int sm2 = static_cast<int>(s) - 2;
#pragma unroll
    for (int g = 0; g < kSlotGroups; g++) {
      const BandSlot sl = S.slot[g * 32 + static_cast<int>(l)];
      const uint32_t a = static_cast<uint32_t>(sl.pad) >> 8, bb = static_cast<uint32_t>(sl.pad) & 255u;
      const bool ok = paired && sl.sab < (1 << 20) && sl.sab > sm2 && a < i && j + 1u + bb < n;
      const uint32_t cls = slot_group_class(static_cast<uint32_t>(g));
      L.g[g] = px[cls * msz + (ok ? o + static_cast<size_t>(static_cast<int64_t>(sl.rel)) : o)];
    }
    uint32_t a = 0, bb = 0;
    if (l < Special<CONTRA>::N) Special<CONTRA>::slot(l, a, bb);
    const bool ok = paired && l < Special<CONTRA>::N && a < i && j + 1u + bb < n && static_cast<int>(a + bb) > sm2;
    const uint32_t k = ok ? i - 1u - a : 0u, ll = ok ? j + 1u + bb : 0u;
    const size_t ok2 = ok ? static_cast<size_t>(k) * ld + ll : 0;
    L.sx = q.m[T_QB][ok2];
    L.spk = q.out[ok ? tri_off(n, ll - k) + k : 0u];
    L.ssc = (l < 4u ? q.m[T_NEAR4] : q.m[T_NEAR8])[4u * ok2 + (l & 3u)];
  };
  auto consume_p = [&](const OutProbes& L, uint32_t s) {
    uint32_t d, j, lim, base, hi;
    bool valid;
    geom(s, d, j, valid, lim, base, hi);
    float* st = S.st[s][c];
    const float qb = st[ST_QB];
    const bool paired = valid && qb > kNegInf;
    const int sm2 = static_cast<int>(s) - 2;
    if (paired && l == 0u) {
      // exterior term (561-573 / 676-680)
      const float qa = st[ST_ACCS], zsj = st[ST_HP];
      st[ST_EXT] = CONTRA ? (((zpi + zsj) + qa) + ext_bp) - ztot : ((zpi + qa) + zsj) - ztot;
    }
    Acc ap = acc_empty();
    if (paired) {
      if (l < Special<CONTRA>::N) {
        uint32_t a, bb;
        Special<CONTRA>::slot(l, a, bb);
        if (a < i && j + 1u + bb < n && static_cast<int>(a + bb) > sm2 && L.sx > kNegInf)
          acc_add(ap, ((L.spk + qb) - L.sx) + L.ssc);
      }
      float xv[kSlotGroups];
#pragma unroll
      for (int g = 0; g < kSlotGroups; g++) {
        const BandSlot sl = S.slot[g * 32 + static_cast<int>(l)];
        const uint32_t a = static_cast<uint32_t>(sl.pad) >> 8, bb = static_cast<uint32_t>(sl.pad) & 255u;
        const bool ok = sl.sab < (1 << 20) && sl.sab > sm2 && a < i && j + 1u + bb < n;
        const uint32_t cls = slot_group_class(static_cast<uint32_t>(g));
        xv[g] = ok ? ((L.g[g] + qb) + sl.len) + st[ST_IN4 + cls] : kNegInf;
      }
#pragma unroll
      for (int g = 0; g + 3 < kSlotGroups; g += 4) acc_add4(ap, xv[g], xv[g + 1], xv[g + 2], xv[g + 3]);
      acc_add(ap, xv[kSlotGroups - 1]);
    }
    ap = half_reduce(ap);
    if (l == 0u) S.pre[s][c][1] = acc_pack(ap);
  };
#ifdef RNAMC_DEBUG_KNOBS
  if (!(b.debug & 128))
#endif
  band_pipeline<OutProbes>(nsteps, issue_p, consume_p);

  struct OutEdges {
    float wa[kEdge], wb[kEdge], la[kEdge], lb[kEdge];
    float2 mid1, mid2;
    float pt;
  };
  auto issue_e = [&](OutEdges& L, uint32_t s) {
    uint32_t d, j, lim, base, hi;
    bool valid;
    geom(s, d, j, valid, lim, base, hi);
    const bool paired = valid && S.st[s < nsteps ? s : 0][c][ST_QB] > kNegInf;
    const float* __restrict__ Wr = w_r + (valid ? row_i + j + 1u : 0);                                  // Wr[x] = W(i, j+1+x)
    const float* __restrict__ Qr = q.m[T_Q1R] + (valid ? static_cast<size_t>(j + 1u) * ld + j : 0);    // Qr[x] = Q1(j+1, j+x)
    const float* __restrict__ A = q.m[T_Q1C] + ((valid && i >= 1u) ? static_cast<size_t>(i - 1u) * ld + 1u : 0);  // A[k] = Q1(k+1, i-1)
    const float* __restrict__ Rc = r_c + (valid ? static_cast<size_t>(j) * ld : 0);                     // Rc[k] = R(k, j)
#pragma unroll
    for (int u = 0; u < kEdge; u++) {
      const uint32_t x = s + l + 32u * u;
      const uint32_t xx = x < lim ? x : 0u;
      L.wa[u] = Wr[xx];
      L.wb[u] = Qr[xx];
      const uint32_t k = base + l + 32u * u;
      const uint32_t kk = (paired && k < hi) ? k : 0u;
      L.la[u] = A[kk];
      L.lb[u] = Rc[kk];
    }
    const uint32_t ring = max(b.ring, 1u);
    L.mid1 = q.mid[(static_cast<size_t>(1u) * b.ring + d % ring) * q.vec + (valid ? i : 0u)];
    L.mid2 = q.mid[(static_cast<size_t>(2u) * b.ring + d % ring) * q.vec + (valid ? i : 0u)];
    // partners of the in-band terms: W(i, j+1+l), l < s; R(i - t, j), t = 2 .. s
    const uint32_t t = l - 8u + 1u;
    const bool pw = valid && l < 8u && l + 1u <= s && l < n - 1u - j;
    const bool pl = paired && l >= 8u && l < 16u && t >= 2u && t <= s && t <= i;
    const float* p = pw ? Qr + l : (pl ? A + (i - t) : q.m[T_QB]);
    L.pt = *p;
  };
  auto consume_e = [&](const OutEdges& L, uint32_t s) {
    uint32_t d, j, lim, base, hi;
    bool valid;
    geom(s, d, j, valid, lim, base, hi);
    const bool paired = valid && S.st[s][c][ST_QB] > kNegInf;
    Acc apm = acc_empty(), ale = acc_empty();
    if (valid) {
      if (thr != 0u && l == 0u && n - 1u - j >= 1u) apm = acc_unpack(L.mid1);
      float x[kEdge];
#pragma unroll
      for (int u = 0; u < kEdge; u++) x[u] = (s + l + 32u * u < lim) ? L.wa[u] + L.wb[u] : kNegInf;
      acc_add4(apm, x[0], x[1], x[2], x[3]);
      acc_add(apm, x[4]);
      if (l < 8u && l + 1u <= s && l < n - 1u - j) S.part[s][c][l] = L.pt;
    }
    if (paired && i >= 1u) {
      if (thr != 0u && l == 0u) ale = acc_unpack(L.mid2);
      float x[kEdge];
#pragma unroll
      for (int u = 0; u < kEdge; u++) x[u] = (base + l + 32u * u < hi) ? L.la[u] + L.lb[u] : kNegInf;
      acc_add4(ale, x[0], x[1], x[2], x[3]);
      acc_add(ale, x[4]);
      const uint32_t t = l - 8u + 1u;
      if (l >= 8u && l < 16u && t >= 2u && t <= s && t <= i) S.part[s][c][l] = L.pt;
    }
    apm = half_reduce(apm);
    ale = half_reduce(ale);
    if (l == 0u) {
      S.pre[s][c][0] = acc_pack(apm);
      S.pre[s][c][2] = acc_pack(ale);
    }
  };
#ifdef RNAMC_DEBUG_KNOBS
  if (!(b.debug & 256))
#endif
  band_pipeline<OutEdges>(nsteps, issue_e, consume_e);
  __syncthreads();

  // ---- phase B: the chain
#ifdef RNAMC_DEBUG_KNOBS
  if (b.debug & 512) return;
#endif
  for (uint32_t s = 0; s < nsteps; s++) {
    const uint32_t d = d0 - s, j = i + d;
    const bool valid = c >= s && ii >= 0 && j < n && d < n;
    if (valid) {
      const float* st = S.st[s][c];
      const float qb = st[ST_QB];
      const bool paired = qb > kNegInf;
      Acc apm = acc_empty(), ap = acc_empty(), ale = acc_empty();
      const uint32_t len0 = n - 1u - j;
      if (l < 8u) {
        if (l + 1u <= s && l < len0) acc_add(apm, IB(s - 1u - l, c)[OUT_W] + S.part[s][c][l]);  // W(i, j+1+l)
      } else if (l < 16u && paired) {
        const uint32_t t = l - 8u + 1u;
        if (t >= 2u && t <= s && t <= i) acc_add(ale, IB(s - t, c - t)[OUT_R] + S.part[s][c][l]);  // R(i-t, j)
      }
      if (paired) {
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
          const uint32_t tt = l + 32u * pass, a = tt >> 3, bb = tt & 7u;
          if (a + bb + 2u <= s && a < i && j + 1u + bb < n) {
            const uint32_t cr = c - 1u - a, sp = s - 2u - a - bb;
            const float pq = IB(sp, cr)[OUT_PQ];  // (log bpp - sums_close) of the closing pair, -inf if none
            float x;
            if (Special<CONTRA>::is(a, bb)) {
              x = (pq + qb) + S.st[sp][cr][ST_NEAR + special_index<CONTRA>(a, bb)];
            } else {
              const uint32_t cls = slot_class(a, bb);
              x = (((pq + S.st[sp][cr][ST_CS4 + cls]) + qb) + S.lenab[a][bb]) + st[ST_IN4 + cls];
            }
            acc_add(ap, x);
          }
        }
      }
      apm = half_reduce(apm);
      acc_merge(apm, acc_unpack(S.pre[s][c][0]));
      const float pm = acc_value(apm);
      const float pm2 = lse2(IB(s - 1, c)[OUT_PM2] + mb_un, IB(s - 1, c)[OUT_W]);
      const float r = lse2(pm, pm2);
      const float sp_prev = IB(s - 1, static_cast<int>(c) - 1)[OUT_SP];  // SP(i-1, j)
      const float spv = lse2(sp_prev + mb_un, pm);
      float lp = kNegInf, w = kNegInf, pq = kNegInf;
      if (paired) {
        ap = half_reduce(ap);
        ale = half_reduce(ale);
        acc_merge(ap, acc_unpack(S.pre[s][c][1]));
        acc_merge(ale, acc_unpack(S.pre[s][c][2]));
        acc_add(ap, st[ST_EXT]);
        const float A = st[ST_ACCS] + abr;
        acc_add(ap, A + acc_value(ale));
        acc_add(ap, A + sp_prev);
        lp = acc_value(ap);
        if (lp > kNegInf) {
          w = (lp + st[ST_MBC]) - qb;
          pq = lp - qb;
        }
      }
      if (l == 0u) {
        float* cell = IB(s, c);
        cell[OUT_LP] = lp;
        cell[OUT_W] = w;
        cell[OUT_PM2] = pm2;
        cell[OUT_R] = r;
        cell[OUT_SP] = spv;
        cell[OUT_PQ] = pq;
      }
      if (c >= kBB - 1) {  // the rows this workgroup owns
        const size_t o = row_i + j, oc = static_cast<size_t>(j) * ld + i;
        if (l == 0u) pm2_r[o] = pm2;
        else if (l == 1u) r_c[oc] = r;
        else if (l == 2u) sp_c[oc] = spv;
        if (lp > kNegInf) {
          if (l == 3u) q.out[tri_off(n, d) + i] = lp;
          else if (l == 4u) q.m[T_ZRE][o] = w;
          else if (l >= 5u && l < 9u) q.m[T_X4][(l - 5u) * msz + o] = pq + st[ST_CS4 + (l - 5u)];
        }
      }
    }
    __syncthreads();
  }
}
#undef IB
