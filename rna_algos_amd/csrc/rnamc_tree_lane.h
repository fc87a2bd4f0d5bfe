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

// (+) of f(x) over x in [lo, hi), four terms a step (loads first, one rescale per step)
template <typename F>
__device__ __forceinline__ void lane_sum(Acc& acc, uint32_t lo, uint32_t hi, F f) {
  uint32_t x = lo;
  for (; x + 4u <= hi; x += 4u) {
    const float v0 = f(x), v1 = f(x + 1u), v2 = f(x + 2u), v3 = f(x + 3u);
    acc_add4(acc, v0, v1, v2, v3);
  }
  if (x < hi) {  // (uniform)
    const float v0 = f(x);
    const float v1 = x + 1u < hi ? f(x + 1u) : kNegInf;
    const float v2 = x + 2u < hi ? f(x + 2u) : kNegInf;
    acc_add4(acc, v0, v1, v2, kNegInf);
  }
}

// The generic 2-loops of one cell: slots e < cnt of the model's list (ordered by a + b), plane of
// the slot's class at diagonal `dbase -/+ (a + b)`, row `i +/- (1 + a)`; st4: the cell's own class
// scores; own: added to every term (outside: sums_close of the cell)
template <bool CONTRA, bool OUTSIDE>
__device__ __forceinline__ void lane_generic(Acc& acc, const TreeBatch& b, const float* __restrict__ x4, size_t msz,
                                             uint32_t ld, uint32_t cnt, uint32_t dbase, uint32_t i, const float4& st4,
                                             float own) {
  const uint32_t* __restrict__ gs = b.tabs->gslot[CONTRA ? 1 : 0];
  const float* __restrict__ gl = b.tabs->glen[CONTRA ? 1 : 0];
  for (uint32_t e = 0; e < cnt; e += 4u) {
    float g[4], ln[4];
    uint32_t cl[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t eu = e + static_cast<uint32_t>(u) < cnt ? e + static_cast<uint32_t>(u) : e;  // (uniform)
      const uint32_t sl = sload(gs + eu);
      ln[u] = sload(gl + eu);
      const uint32_t a = sl & 31u, bb = (sl >> 5) & 31u;
      cl[u] = sl >> 10;
      const size_t o = OUTSIDE ? static_cast<size_t>(dbase + a + bb) * ld + (i - 1u - a)
                               : static_cast<size_t>(dbase - a - bb) * ld + (i + 1u + a);
      g[u] = x4[cl[u] * msz + o];
    }
    float x[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const float v = OUTSIDE ? ((g[u] + own) + ln[u]) + pick(st4, cl[u]) : (g[u] + ln[u]) + pick(st4, cl[u]);
      x[u] = e + static_cast<uint32_t>(u) < cnt ? v : kNegInf;
    }
    acc_add4(acc, x[0], x[1], x[2], x[3]);
  }
}

// ---- inside sweep, diagonal d (src/mccaskill_algo.rs:296-351 / 430-486)
template <bool CONTRA>
__global__ void __launch_bounds__(256) k_tlane_inside(TreeBatch b, uint32_t d, uint32_t thr) {
  const TSeq q = load_tseq(b, blockIdx.y);
  const uint32_t n = q.n, ld = q.ld;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i + d >= n) return;
  const uint32_t j = i + d;
  const size_t msz = static_cast<size_t>(q.m[1] - q.m[0]);
  const size_t row = static_cast<size_t>(i) * ld + j, col = static_cast<size_t>(j) * ld + i;
  const size_t dg = static_cast<size_t>(d) * ld + i;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // closing-pair block
  const float mbc = q.m[T_MBC][row];
  float qa = kNegInf;
  if (mbc > kNegInf) {
    const float hp = q.m[T_HP][row];
    const float4 cs = reinterpret_cast<const float4*>(q.m[T_CS4])[row];
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
    Acc acc = acc_empty();
    acc_add4(acc, hp, qm + mbc, xs[0], xs[1]);
    if (CONTRA) {
      acc_add2(acc, xs[2], xs[3]);
    } else {
      acc_add4(acc, xs[2], xs[3], xs[4], xs[5]);
      acc_add(acc, xs[6]);
    }
    if (d >= 5u) {  // (a generic slot has a + b >= 2)
      const uint32_t cnt = sload(&b.tabs->gcount[CONTRA ? 1 : 0][min(d - 3u, 30u)]);
      lane_generic<CONTRA, false>(acc, b, q.m[T_X4], msz, ld, cnt, d - 2u, i, cs, 0.f);
    }
    const float qb = acc_value(acc);
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
  // sums_multibranch: x = Q1's span, Q1(i, i+x) + Zr_mb(i+1+x, j); banded (thr != 0): the terms with
  // both spans below thr are k_tree_mid's
  Acc pm = acc_empty();
  if (d >= 2u) {
    const float* __restrict__ A = q.m[T_Q1_D] + i;
    const float* __restrict__ B = q.m[T_ZRM_D] + (i + 1u);
    auto term = [&](uint32_t x) { return A[static_cast<size_t>(x) * ld] + B[static_cast<size_t>(d - 1u - x) * ld + x]; };
    if (thr != 0u) {
      lane_sum(pm, 0u, d - thr, term);
      lane_sum(pm, thr, d - 1u, term);
      const float2 mm = q.mid[static_cast<size_t>(d % b.ring) * q.vec + i];
      acc_merge(pm, Acc{mm.x, mm.y});
    } else {
      lane_sum(pm, 0u, d - 1u, term);
    }
  }
  const float qmv = acc_value(pm);
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

}  // namespace

void launch_tlane_inside(const TreeBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq, uint32_t thr,
                         hipStream_t st) {
  const uint32_t gx = (max_n - d + 255u) / 256u;
  if (contra)
    hipLaunchKernelGGL(k_tlane_inside<true>, dim3(gx, nseq, 1), dim3(256), 0, st, b, d, thr);
  else
    hipLaunchKernelGGL(k_tlane_inside<false>, dim3(gx, nseq, 1), dim3(256), 0, st, b, d, thr);
}
