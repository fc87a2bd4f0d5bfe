// rnamc_latency.h — latency forms of the folds, for lock-step groups too small to fill the
// chip (a single long sequence: BASELINE config 3, n = 4096).  Device code, included by
// rnamc_kernels.hip inside namespace rnamc { namespace { ... } }.
//
// Such a group is bound by the LENGTH of the dependent fold chains the reference's summation
// order dictates (src/mccaskill_algo.rs:344-374 / 468-512 inside: n^2/2 steps; 594-601 /
// 701-714 outside: 3 n^2/2 steps), not by HBM or VALU throughput: > 95 % of the chip idles,
// and a lone wave issues one instruction every ~6.7 cycles whatever its kind
// (scripts/ubench/far_step.hip).  Every form here is the same operations on the chain's
// value in the same order, bit for bit, laid out for the fewest instructions per step:
//  * outside chains and 2-loop rows, one wave per chain (fold_block): the terms of the next
//    block are formed lane-parallel and classified AHEAD of the chain; a term certain to meet
//    the identity piece of logsumexp (`min + (max - min)`, src/utils.rs:589-591: 96 % of the
//    outside pair-probability steps) costs two VALU instructions, the running sum walking
//    from lane to lane through a DPP operand; the others take lse_w — a scalar branch for the
//    identity piece, else lane p (mod 8) evaluates cubic piece p of ln_exp_1p
//    (src/utils.rs:602-627) and a DPP OR over each group of 8 lanes hands the selected
//    result to every lane: no LDS table on the chain;
//  * inside folds, eight chains per wave (chain_e): inside terms stay near the sum, so every
//    step evaluates the cubic — lanes 8g..8g+7 hold one cell's chain, one piece per lane;
//  * operands are loaded lane-distributed, blocks ahead of the chain (a step takes 6-90 ns,
//    a round trip to L2 / HBM 1-2 us), and handed on through opaque moves so that the
//    compiler waits for exactly the load it needs (counted vmcnt).
#ifndef RNAMC_LATENCY_H
#define RNAMC_LATENCY_H

struct Piece8 {
  float c0, c1, c2, c3;  // cubic piece (threadIdx & 7)
  float tlo, thi;        // its interval [tlo, thi)
  // the same interval on the bits of z (z >= +0: bit order is value order), piece 7's
  // stretched to every finite z above its lower end:  (bits(z) - lo_bits) < width
  uint32_t lo_bits, width;
};

__device__ __forceinline__ Piece8 load_piece8() {
  const int p = static_cast<int>(threadIdx.x & 7u);
  Piece8 r;
  r.c0 = kLseCoef[p][0];
  r.c1 = kLseCoef[p][1];
  r.c2 = kLseCoef[p][2];
  r.c3 = kLseCoef[p][3];
  r.tlo = p ? kLseBreaks[p - 1] : -1.f;  // z >= 0 always
  r.thi = kLseBreaks[p];
  r.lo_bits = p ? __float_as_uint(kLseBreaks[p - 1]) : 0u;
  r.width = (p == 7 ? 0x7F800000u : __float_as_uint(kLseBreaks[p])) - r.lo_bits;
  return r;
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_or(uint32_t v) {
  return v | static_cast<uint32_t>(
                 __builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, false));
}

// Operands: lane l of the wave loads step k0 + l (one coalesced load per 64 steps and
// stream), the chain takes them lane by lane through v_readlane: the next 64-128 steps are in
// flight in four registers, far beyond any memory latency.
// A loaded register handed on through an opaque move: the compiler waits for exactly that
// load HERE (a counted vmcnt in straight-line code) instead of draining every load in flight
// at the top of the loop that consumes it.
__device__ __forceinline__ float pin(float v) {
  float o;
  asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(v));
  return o;
}

__device__ __forceinline__ float lane_val(float v, uint32_t lane) {
  return __uint_as_float(static_cast<uint32_t>(
      __builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(v)), static_cast<int>(lane))));
}

constexpr uint32_t kLseThrBits = 0x413DCCB7u;  // 11.862479f (LOGSUMEXP_THRESHOLD_UPPER)

// One fold step sum ⊕ x of the wave's chain: `sum` and `x` are the same in every lane.
__device__ __forceinline__ float lse_u(float sum, float x, const Piece8& P) {
  const float hi = vmax(sum, x);
  const float lo = vmin(sum, x);
  const float z = hi - lo;  // >= 0; +inf or NaN when lo is -inf
  // 11.862479 <= z < +inf as one unsigned compare on the bits (z >= 0, NaN sorts above inf)
  const bool far = (__float_as_uint(z) - kLseThrBits) < (0x7F800000u - kLseThrBits);
  if (__builtin_expect(__ballot(far) != 0ull, 1)) return lo + z;  // identity piece, finite operands
  // z < 11.862479: the 8 cubic pieces, one per lane (mod 8); or lo = -inf.  The lanes whose
  // interval holds z (one per group of 8, all with the same value) are found by a ballot,
  // the first of them hands its result to the whole wave through an SGPR.
  float r = ((P.c0 * z + P.c1) * z + P.c2) * z + P.c3;
  r = lo + r;
  const unsigned long long sel = __ballot((z >= P.tlo) && (z < P.thi));
  if (sel == 0ull) return hi;  // z is +inf or NaN: lo is -inf, the sum is hi
  return lane_val(r, static_cast<uint32_t>(__builtin_ctzll(sel)));
}

__device__ __forceinline__ uint32_t uni(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
}

// One general fold step of a wave-uniform chain with FINITE operands (`sum` comes out of a
// lane read, i.e. an SGPR): the identity piece is a scalar branch away; otherwise lane p
// (mod 8) evaluates cubic piece p and a 3-step DPP OR over each group of 8 hands the result of
// the lane whose interval holds z to every lane (all groups see the same operands) — no
// ballot, no second lane read.
__device__ __forceinline__ float lse_w(float sum, float x, const Piece8& P) {
  float hi, lo;
  asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "s"(sum), "v"(x));
  asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "s"(sum), "v"(x));
  const float z = hi - lo;  // finite, >= 0
  if (__ballot(__float_as_uint(z) >= kLseThrBits) != 0ull) return lo + z;
  float r = ((P.c0 * z + P.c1) * z + P.c2) * z + P.c3;
  r = lo + r;
  const bool sel = (__float_as_uint(z) - P.lo_bits) < P.width;
  uint32_t v = sel ? __float_as_uint(r) : 0u;
  v = dpp_or<0xB1>(v);
  v = dpp_or<0x4E>(v);
  v = dpp_or<0x141>(v);
  return __uint_as_float(v);
}

// Terms classified AHEAD of the chain.  The terms of the next block of steps (at most 256)
// are computed lane-parallel; a finite term t < far_limit(sum at block start, ...) is certain
// to meet the identity piece when the chain reaches it, whatever the steps in between did:
// a fold step never lowers the running sum by more than its two roundings (every cubic piece
// returns more than z).  With M >= every |term| of the block and >= |sum| + 256 (the sum
// gains at most ln 2 per step), z < 2 M, one step loses < 1.5 M 2^-23 and 256 steps
// < M 2^-14.4: the margin kept in hand is 1 for M < 2^12, 4 below 2^16, 64 below 2^20, and
// beyond that nothing is classified.  For such a step logsumexp IS  t + (sum - t):  hi = sum,
// lo = t, z = hi - lo >= 11.862479, result lo + z (src/utils.rs:589-591) — two dependent
// instructions (far_steps), no compare and no branch on the chain.  A term of -inf leaves the
// sum as it is (lse_u returns hi) and is no step at all.
// `mag`: this lane's largest finite |term| (0 if it has none).
__device__ __forceinline__ float far_limit(float sum, float mag) {
  const float m = vmax(mag, __builtin_fabsf(sum) + 256.f);  // (sum = -inf: nothing classified)
  const bool b12 = __builtin_amdgcn_ballot_w64(m >= 4096.f) != 0ull;
  const bool b16 = __builtin_amdgcn_ballot_w64(m >= 65536.f) != 0ull;
  const bool b20 = __builtin_amdgcn_ballot_w64(m >= 1048576.f) != 0ull;
  const float margin = !b12 ? 1.f : !b16 ? 4.f : !b20 ? 64.f : __builtin_inff();
  return sum - (11.862479f + margin);
}
__device__ __forceinline__ bool sure_far(float t, float lim) { return t < lim && t > kNegInf; }

__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {  // set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
}

// N sure-far steps of the walking sum (fold_block): lane l takes the sum of lane l-1 through
// the DPP operand of its subtraction.
template <int N>
__device__ __forceinline__ float far_steps(float s, float t) {
#pragma unroll
  for (int u = 0; u < N; u++) {
    const float prev = __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_update_dpp(
        0, static_cast<int>(__float_as_uint(s)), 0x13C /* wave_ror:1 */, 0xF, 0xF, false)));
    const float z = prev - t;
    s = t + z;
  }
  return s;
}

// One block of a chain whose steps come NT per lane (lane l: steps NT*l .. NT*l + NT-1, a
// term of -inf = no step).  The finite terms are compacted in step order through LDS
// (`buf`: 64 * NT floats of this wave's own), classified against the sum at block start,
// and the chain then alternates between runs of sure-far steps and single general steps.
template <int NT>
__device__ __forceinline__ float fold_block(float sum, const float (&a)[NT], float* buf,
                                            const Piece8& P8) {
  // far_limit's margins are derived for blocks of at most 256 steps (tests/test_oracle_cpu.py
  // checks the per-piece bound r(z) >= z - eps they rest on)
  static_assert(NT * 64 <= 256, "fold_block: the sure-far margins cover at most 256 steps per block");
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t pos = 0, total = 0;
  float mag = 0.f;
#pragma unroll
  for (int c = 0; c < NT; c++) {
    const unsigned long long present = __builtin_amdgcn_ballot_w64(a[c] > kNegInf);
    pos += lanes_below(present);
    total += static_cast<uint32_t>(__popcll(present));
    if (a[c] > kNegInf) mag = vmax(mag, __builtin_fabsf(a[c]));
  }
  if (total == 0u) return sum;
  __builtin_amdgcn_wave_barrier();  // (the previous block's reads are done)
#pragma unroll
  for (int c = 0; c < NT; c++) {
    if (a[c] > kNegInf) buf[pos++] = a[c];
  }
  __builtin_amdgcn_wave_barrier();
  // -inf (+) x is x (lse_u returns hi): a chain that has not met a finite term yet takes the
  // block's first one as it is, and every step from there on has FINITE operands
  uint32_t first = 0;
  if (uni(__float_as_uint(sum)) == 0xFF800000u) {
    sum = buf[0];
    first = 1u;
  }
  const float lim = far_limit(sum, mag);
  // A lone wave issues one instruction every ~6.7 cycles whatever its kind
  // (scripts/ubench/far_step.hip), so the chain is laid out for the fewest instructions per
  // step: the running sum WALKS across the lanes.  Step l happens in lane l, which takes the
  // sum from lane l-1 through the DPP operand of its subtraction (wave_ror:1; lane 0 reads
  // lane 63, where the block's incoming sum still stands) — a sure-far step is two VALU
  // instructions and no lane read.  The other lanes compute values nobody uses.  A general
  // step fetches sum and term into SGPRs, runs lse_w and leaves its result in every lane.
  for (uint32_t w0 = 0; w0 < total; w0 += 64u) {
    const uint32_t cnt = min(64u, total - w0);
    const float t = buf[w0 + lane];  // (past `total`: stale, masked below)
    const unsigned long long far = __builtin_amdgcn_ballot_w64(lane < cnt && sure_far(t, lim));
    float s = sum;  // uniform here
    uint32_t l = w0 == 0u ? first : 0u;
    while (l < cnt) {
      // the run of sure-far steps from l on (bits past cnt are clear), in straight-line
      // pieces of 16, 8, 4, 2, 1 steps: ~2 instructions per step and a dozen per run
      const unsigned long long stop = ~(far >> l);
      uint32_t r = stop ? static_cast<uint32_t>(__builtin_ctzll(stop)) : 64u;
      l += r;
      for (; r >= 16u; r -= 16u) s = far_steps<16>(s, t);
      if (r & 8u) s = far_steps<8>(s, t);
      if (r & 4u) s = far_steps<4>(s, t);
      if (r & 2u) s = far_steps<2>(s, t);
      if (r & 1u) s = far_steps<1>(s, t);
      if (l < cnt) {
        s = lse_w(lane_val(s, (l + 63u) & 63u), lane_val(t, l), P8);
        l++;
      }
    }
    sum = lane_val(s, cnt - 1u);
  }
  return sum;
}


// ----------------------------------------------------------------------------
// inside, eight chains per wave: lanes 8g..8g+7 hold the chain of cell cell0 + g, one role per wave
//   role 0: sums_rightmost_basepairs_external (CONTRAfold: its own fold) and sums_external
//   role 1: (CONTRAfold: sums_rightmost_basepairs_multibranch and) the first sum of 364-374 /
//           499-512, parked in the sums_1ormore slot until inside_combine_lat
//   role 2: sums_multibranch
// (same operations per chain as inside_sums_cell); lane p of a group evaluates cubic piece p and an OR over the group (3 DPP
// steps) picks the piece whose interval holds z: the LDS-table round trips (60 % of a
// dependent step of the three-lanes-per-cell form) leave the chain.  No scalar fast path here:
// eight independent chains are hardly ever all in the identity piece at once.
__device__ __forceinline__ float lse8(float sum, float x, const Piece8& P) {
  const float hi = vmax(sum, x);
  const float lo = vmin(sum, x);
  const float z = hi - lo;  // >= 0; +inf or NaN when lo is -inf
  float r = ((P.c0 * z + P.c1) * z + P.c2) * z + P.c3;
  r = lo + r;
  const bool sel = (z >= P.tlo) && (z < P.thi);
  uint32_t v = sel ? __float_as_uint(r) : 0u;
  v = dpp_or<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_or<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_or<0x141>(v);  // row_half_mirror: the other quad of the 8
  // z >= 11.862479: identity piece lo + z; lo = -inf (z +inf or NaN): the sum is hi
  const float alt = (lo == kNegInf) ? hi : lo + z;
  return (z < 11.862479f) ? __uint_as_float(v) : alt;
}

// The same step when both operands are FINITE (so z is): the lane of piece 7 also serves
// the identity piece — it adds lo to z itself instead of to its cubic when z >= 11.862479
// (src/utils.rs:589-591) — and the OR over the group is the result: 18 instructions where
// lse8 takes 22.
__device__ __forceinline__ float lse8_fin(float sum, float x, const Piece8& P) {
  const float hi = vmax(sum, x);
  const float lo = vmin(sum, x);
  const float z = hi - lo;
  float r = ((P.c0 * z + P.c1) * z + P.c2) * z + P.c3;
  r = (z >= 11.862479f) ? z : r;
  r = lo + r;
  const bool sel = (__float_as_uint(z) - P.lo_bits) < P.width;
  uint32_t v = sel ? __float_as_uint(r) : 0u;
  v = dpp_or<0xB1>(v);
  v = dpp_or<0x4E>(v);
  v = dpp_or<0x141>(v);
  return __uint_as_float(v);
}

// The value lane 8g + U holds, in every lane of group g (two DPP moves: each quad spreads
// its lane U & 3, then the quads that do not hold lane U take the other quad's through
// row_half_mirror).
template <int U>
__device__ __forceinline__ float bcast8(float v) {
  constexpr int q = U & 3;
  const int a = __builtin_amdgcn_update_dpp(0, static_cast<int>(__float_as_uint(v)),
                                            q | (q << 2) | (q << 4) | (q << 6), 0xF, 0xF, false);
  const int b = __builtin_amdgcn_update_dpp(a, a, 0x141 /* row_half_mirror */, 0xF,
                                            (U < 4) ? 0xA : 0x5, false);
  return __uint_as_float(static_cast<uint32_t>(b));
}

// Steps t_first .. t_last of eight chains (one per group of eight lanes).  The terms are
// formed LANE-PARALLEL — lane p of a group loads the operands of step t0 + p (load), four such
// blocks per operand buffer, and turns them into the term (form) — and reach the group
// through bcast8: per step two DPP moves instead of the loads, address arithmetic and term
// arithmetic of every lane for itself.  A block of eight steps whose terms and running sums
// are all finite takes lse8_fin.
constexpr uint32_t kEB = 32;  // steps per operand buffer
struct EOp {
  float a, b;  // the (up to) two operands of a step, as loaded
};
template <class Load, class Form>
__device__ __forceinline__ float chain_e(float acc, uint32_t t_first, uint32_t t_last, Load&& load,
                                         Form&& form, const Piece8& P8) {
  if (t_last < t_first) return acc;
  const uint32_t p = threadIdx.x & 7u;
  struct GBuf {
    EOp o[4];
  };
  (void)pingpong<GBuf, kEB, true>(
      t_first, (t_last - t_first + kEB) / kEB,
      [&](GBuf& B, uint32_t t0) {
#pragma unroll
        for (int k = 0; k < 4; k++) B.o[k] = load(min(t0 + 8u * k + p, t_last));
      },
      [&](const GBuf& B, uint32_t t0) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t tk = t0 + 8u * k;
          if (tk > t_last) break;
          const EOp o = {pin(B.o[k].a), pin(B.o[k].b)};
          const float g = form(o, min(tk + p, t_last));
          const uint32_t cnt = min(8u, t_last + 1u - tk);
          const bool fin = __builtin_amdgcn_ballot_w64(g == kNegInf || acc == kNegInf) == 0ull;
          if (cnt == 8u && fin) {
            acc = lse8_fin(acc, bcast8<0>(g), P8);
            acc = lse8_fin(acc, bcast8<1>(g), P8);
            acc = lse8_fin(acc, bcast8<2>(g), P8);
            acc = lse8_fin(acc, bcast8<3>(g), P8);
            acc = lse8_fin(acc, bcast8<4>(g), P8);
            acc = lse8_fin(acc, bcast8<5>(g), P8);
            acc = lse8_fin(acc, bcast8<6>(g), P8);
            acc = lse8_fin(acc, bcast8<7>(g), P8);
          } else {
            acc = lse8(acc, bcast8<0>(g), P8);
            if (cnt > 1u) acc = lse8(acc, bcast8<1>(g), P8);
            if (cnt > 2u) acc = lse8(acc, bcast8<2>(g), P8);
            if (cnt > 3u) acc = lse8(acc, bcast8<3>(g), P8);
            if (cnt > 4u) acc = lse8(acc, bcast8<4>(g), P8);
            if (cnt > 5u) acc = lse8(acc, bcast8<5>(g), P8);
            if (cnt > 6u) acc = lse8(acc, bcast8<6>(g), P8);
            if (cnt > 7u) acc = lse8(acc, bcast8<7>(g), P8);
          }
        }
      });
  return acc;
}

// (ROLE is a template parameter: chosen per step at run time, the wave-uniform role conditions
// became a handful of scalar branches in every fold step)
template <bool CONTRA, uint32_t ROLE>
__device__ __forceinline__ void inside_chain_e(const DeviceBatch& b, const Seq& q, uint32_t d,
                                               uint32_t cell0, const Piece8& P8, bool zr_parked) {
  constexpr uint32_t role = ROLE;
  const uint32_t n = q.n;
  const uint32_t lane = threadIdx.x & 63u;
  // roles 3, 4 (CONTRAfold): all but the last step of sums_rightmost_basepairs_{external,
  // multibranch} of diagonal d + 1 — they need sums_accessible of diagonals <= d only, so the
  // two dependent chains of a CONTRAfold cell (468-486, then 487-512) overlap across launches
  constexpr bool AHEAD = ROLE >= 3;
  const uint32_t D = AHEAD ? d + 1u : d;  // the diagonal this wave's cells lie on
  if (D >= n) return;
  const uint32_t cells = n - D;
  const uint32_t ic = cell0 + (lane >> 3);
  if (cell0 >= cells) return;
  const bool valid = ic < cells;
  const uint32_t i = valid ? ic : cells - 1u;  // groups past the diagonal shadow its last cell
  const bool leader = valid && (lane & 7u) == 0u;
  const uint32_t od = tri_off(n, D) + i;
  const float* __restrict__ zre = q.m[M_ZRE];
  const float* __restrict__ zrm = q.m[CONTRA ? M_ZRM : M_ZRE];
  const float* __restrict__ qa = q.m[M_QA];
  float zr = kNegInf, c = 0.f, mun = 0.f;
  if (!CONTRA) {
    const float prev = (d >= 1) ? zre[tri_off(n, d - 1) + i] : kNegInf;
    zr = lse8(prev, qa[od], P8);
    if (leader && role == 0) q.m[M_ZRE][od] = zr;
    c = b.params->turner.coeff_num_branches;
  } else {
    const rnamc_fold_score_sets& f = b.params->contra;
    mun = f.multibranch_score_unpair;
    if (role != 2) {
      const bool ext = role == 0 || role == 3;
      const float Pc = ext ? f.external_score_basepair : f.multibranch_score_basepair;
      const float Qc = ext ? f.external_score_unpair : mun;
      float* __restrict__ slot = q.m[ext ? M_ZRE : M_ZRM] + od;
      if (!AHEAD && zr_parked) {
        // steps 1 .. d-1 were folded beside the previous diagonal: the last one is left
        zr = lse8(*slot, qa[od] + Pc + Qc * 0.f, P8);
      } else {
        const uint32_t last = AHEAD ? D - 1u : D;  // steps t = 1 .. last of the D-step fold
        zr = chain_e(
            zr, 1u, last, [&](uint32_t t) { return EOp{qa[tri_off(n, t) + i], 0.f}; },
            [&](const EOp& o, uint32_t t) { return o.a + Pc + Qc * static_cast<float>(D - t); }, P8);
      }
      if (leader) *slot = zr;  // (AHEAD: parked partial sum)
      if (AHEAD) return;
    }
  }
  float acc;
  if (role == 0) {
    acc = CONTRA ? lse8(b.params->contra.external_score_unpair * static_cast<float>(d + 1), zr + 0.f, P8)
                 : lse8(0.f, zr + 0.f, P8);  // k = i: Z[i][i-1] is the lower-triangle 0
  } else if (role == 1) {
    acc = CONTRA ? zr : zr + c;
  } else {
    acc = kNegInf;
  }
  const float* __restrict__ pa = (CONTRA && role != 0) ? zrm : zre;
  const float* __restrict__ pb = q.m[role == 0 ? M_Z : M_Q1D];
  if (d >= 2u) {
    constexpr bool one = role == 1;  // (its terms have one operand)
    acc = chain_e(
        acc, 1u, d - 1u,
        [&](uint32_t t) {
          return EOp{pa[tri_off(n, d - t) + t + i], one ? 0.f : pb[tri_off(n, t - 1u) + i]};
        },
        [&](const EOp& o, uint32_t t) {
          if (!CONTRA) return (role == 0) ? o.a + o.b : (role == 1 ? o.a + c : o.b + (o.a + c));
          return (role == 1) ? o.a + mun * static_cast<float>(t) : o.b + o.a;
        },
        P8);
  }
  if (leader) {
    if (role == 0) q.m[M_Z][od] = acc;
    if (role == 1) q.m[M_Q1D][od] = acc;  // parked: inside_combine_lat turns it into sums_1ormore
    if (role == 2) q.m[M_QM][od] = acc;
  }
}

// sums_1ormore_basepairs(i,j) = parked first sum ⊕ sums_multibranch (374 / 512), both layouts.
// One lane per cell; runs in the launch of the NEXT diagonal (whose folds read nothing newer
// than diagonal d-1 of this matrix).
__device__ __forceinline__ void inside_combine_lat(const Seq& q, uint32_t d, uint32_t i,
                                                   const LseTab* tab) {
  const uint32_t od = tri_off(q.n, d) + i;
  const float q1v = lse(q.m[M_Q1D][od], q.m[M_QM][od], tab);
  q.m[M_Q1D][od] = q1v;
  if (i >= 1) q.m[M_Q1C][col_off(i + d) + i - 1] = q1v;
}

// ----------------------------------------------------------------------------
// outside, probs_multibranch{,2} of one cell per wave (src/mccaskill_algo.rs:540-557 /
// 641-661).  Only the partners k of base i are folded (scalar scan of the 2-bit packed
// sequence); an absent pair is a map miss in the reference too.  In a latency-form group
// W = (P + mbclose) - Qb is stored DENSE (by position, not by list index: every launch of
// the group uses these forms).
template <bool CONTRA>
__device__ __forceinline__ void outside_mb_lat(const DeviceBatch& b, const Seq& q, uint32_t d,
                                               uint32_t i, const Piece8& P8) {
  const uint32_t n = q.n;
  const uint32_t j = i + d;
  const uint32_t lane = threadIdx.x & 63u;
  const float* __restrict__ q1d = q.m[M_Q1D];
  const float* __restrict__ w = q.m[M_W];
  const uint8_t* __restrict__ s = q.s;
  const float mun = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  const uint32_t cnt = n - 1u - j;  // steps t = 1 .. cnt, k = j + t
  // partners of base a as a 4-bit set: A:{U} C:{G} G:{C,U} U:{A,G}
  const uint32_t pairmask = (0x5A48u >> (4u * s[i])) & 15u;
  float pm = kNegInf, pm2 = kNegInf;
  __shared__ float steps[64];
  // Blocks of 64 steps: lane l owns step t0 + l — it loads that step's two operands and
  // decides whether base k pairs with base i; a ballot gives the block's partner set, whose
  // members the chain then visits in ascending k through scalar bit scans and v_readlane.
  struct MBuf {
    float x, r;
    uint32_t has;
  };
  (void)pingpong<MBuf, 64, true>(
      1u, (cnt + 63u) / 64u,
      [&](MBuf& B, uint32_t t0) {
        const uint32_t t = t0 + lane;
        const bool in = t <= cnt;
        const uint32_t tt = in ? t : cnt;  // (cnt >= 1 here)
        B.x = w[tri_off(n, d + tt) + i];
        B.r = q1d[tri_off(n, tt >= 2u ? tt - 2u : 0u) + d + 1u + i];
        B.has = (in && ((pairmask >> s[j + tt]) & 1u)) ? 1u : 0u;
      },
      [&](const MBuf& B, uint32_t t0) {
        const float bx = pin(B.x), br = pin(B.r);
        // both chains' terms of step t0 + lane, classified against the sums at block start
        const float a1 = bx + br;
        const float a2 = CONTRA ? bx + mun * static_cast<float>(t0 + lane - 1u) : bx;
        const bool has1 = B.has != 0u && t0 + lane >= 2u;  // (t = 1: only pm2 moves)
        const float t1[1] = {has1 ? a1 : kNegInf};
        const float t2[1] = {B.has != 0u ? a2 : kNegInf};
        pm = fold_block<1>(pm, t1, steps, P8);
        pm2 = fold_block<1>(pm2, t2, steps, P8);
      });
  if (lane == 0u) reinterpret_cast<float2*>(q.m[M_PM])[col_off(j) + i] = make_float2(pm, pm2);
}

// ----------------------------------------------------------------------------
// outside, multibranch half of the pair probability of one listed cell per wave
// (src/mccaskill_algo.rs:594-605 / 701-718): ONE chain of three fold steps per k.
template <bool CONTRA>
__device__ __forceinline__ void outside_tail_lat(const DeviceBatch& b, const Seq& q, uint32_t d,
                                                 uint32_t i, const Piece8& P8) {
  const uint32_t n = q.n;
  const uint32_t od = tri_off(n, d) + i;
  const float qb_ij = q.m[M_QB][od];
  if (uni(__float_as_uint(qb_ij)) == 0xFF800000u) return;  // no pair (CONTRAfold), uniform
  float p = q.m[M_P][od];  // the 2-loop half parked by outside_pair_head
  const float qa_ij = q.m[M_QA][od];
  const float sa = CONTRA ? qa_ij + b.params->contra.multibranch_score_basepair
                          : qa_ij + b.params->turner.coeff_num_branches;
  const float mun = CONTRA ? b.params->contra.multibranch_score_unpair : 0.f;
  const uint32_t j = i + d;
  const uint32_t lane = threadIdx.x & 63u;
  // column j of {probs_multibranch, probs_multibranch2}; column i-1 of sums_1ormore, one row
  // up: x_k = sums_1ormore_basepairs[k+1][i-1], the empty interval (-inf) for k = i-1
  const float2* __restrict__ yycol = reinterpret_cast<const float2*>(q.m[M_PM]) + col_off(j);
  const float* __restrict__ xcol = q.m[M_Q1C] + col_off(i >= 1 ? i - 1 : 0);
  const uint32_t full = i >= 1 ? i - 1 : 0;  // steps k < full have all three terms
  struct TBuf {
    float x;
    float2 yy;
  };
  __shared__ float steps[192];
  // whole blocks of 64; a read past the cell's own rows stays inside the padded matrices
  // (always-fetch form: the number of loads in flight at the wait is static, so the wait for
  // this block's operands leaves the next block's loads in flight)
  (void)pingpong<TBuf, 64, true>(
      0u, (full + 63u) / 64u,
      [&](TBuf& B, uint32_t k0) {
        B.x = xcol[k0 + lane];
        B.yy = yycol[k0 + lane];
      },
      [&](const TBuf& B, uint32_t k0) {
        const uint32_t cnt = min(64u, full - k0);
        const float bx = pin(B.x), by = pin(B.yy.x), by2 = pin(B.yy.y);
        // the three terms of k = k0 + lane, classified against p at block start
        const float a0 = sa + by2 + bx;
        const float a1 = CONTRA ? sa + by + mun * static_cast<float>(i - (k0 + lane) - 1) : sa + by;
        const float a2 = sa + bx + by;
        const bool in = lane < cnt;
        const float a[3] = {in ? a0 : kNegInf, in ? a1 : kNegInf, in ? a2 : kNegInf};
        p = fold_block<3>(p, a, steps, P8);
      });
  if (i >= 1) {
    // k = i - 1: the interval [k+1, i-1] is empty, only the middle term exists
    const float2 yy = yycol[i - 1];
    p = lse_u(p, CONTRA ? sa + yy.x + mun * 0.f : sa + yy.x, P8);
  }
  if (lane == 0u && p > kNegInf) {
    q.m[M_P][od] = p;
    reinterpret_cast<float2*>(q.m[M_PQ])[od] = make_float2(p, qb_ij);
    q.m[M_W][od] = p + q.m[M_MBC][od] - qb_ij;  // dense: read by outside_mb_lat
  }
}

// ----------------------------------------------------------------------------
// Turner 2-loop score with every table lookup issued at once (get_2loop_score,
// src/utils.rs:207-366; same expression tree per loop class as Turner::twoloop of
// rnamc_scoring.h, which walks the classes through branches: lanes of one wave hold different
// classes, so the branchy form pays the dependent global loads of every class in turn,
// ~1.8 us per call).  The bases come from two 40-byte windows staged in LDS (Win2), the
// class picks ONE primary table entry per lane (stack / 1x1 / 1x2 / 2x2 / the terminal
// mismatch table of the class), the second mismatch entry and the initiation entry are
// loaded beside it: one round trip per call.
// (ci,cj) closes, (ak,al) is enclosed; x1,x2 = the two bases inside ci, y1,y2 inside cj;
// m2,m3 = the bases outside al and ak; a, b = unpaired bases on the two sides.
struct Win2 {
  uint8_t left[40], right[40];
};

// (turner_twoloop_flat itself lives in rnamc_scoring.h: the tree-order kernels use it too)

// ----------------------------------------------------------------------------
// The 2-loop blocks of one listed cell per wave.  The probes (a, b) are taken in fold order
// (a ascending, then b), 64 per batch; lane l loads the operand(s) of probe m0 + l and scores
// the loop with the plain scorer of rnamc_scoring.h (same expression trees as the
// table-driven one of rnamc_probes.h; this is the scorer behind FoldScores), and fold_block
// folds the batch's PRESENT pairs in lane order.  An absent pair is a map miss in the reference
// (no fold step at all).  The next batch's terms are computed before the current one is folded.
template <bool CONTRA, int MODE>
__device__ __forceinline__ void inside_pair_lat(const DeviceBatch& b, const Seq& q, uint32_t d,
                                                uint32_t i, const Piece8& P8) {
  const uint32_t n = q.n;
  const uint32_t j = i + d;
  const uint8_t* __restrict__ s = q.s;
  const uint32_t lane = threadIdx.x & 63u;
  if (!(b.allows_short_hairpins && CONTRA) && d + 1 < RNAMC_MIN_SPAN_HAIRPIN_CLOSE) return;
  const auto model = ModelOf<CONTRA>::make(b);
  const uint32_t o = tri_off(n, d) + i;
  const float* __restrict__ qb = q.m[M_QB];
  __shared__ float steps[64];
  float sum = kNegInf;
  if (!CONTRA || d - 1 <= RNAMC_MAX_LOOP_LEN) sum = lse_u(sum, model.hairpin(s, n, i, j), P8);
  if (d >= 3) {
    // enclosed pairs (k,l) = (i+1+a, j-1-bb), a ascending, bb ascending (l descending),
    // a + bb <= 30, l > k  <=>  a + bb <= d-3
    const uint32_t lim = min(static_cast<uint32_t>(RNAMC_MAX_2LOOP_LEN), d - 3);
    // the (lim+1)(lim+2)/2 probes in fold order, 64 per batch: lane l scores probe m0 + l
    // (a scorer call costs ~1.8 us of dependent table lookups however many lanes take part:
    // 8 batches instead of 31 rows)
    const uint32_t nprobe = (lim + 1u) * (lim + 2u) / 2u;
    // the bases around i (rightwards) and j (leftwards), for the flat scorers
    __shared__ Win2 win;
    if (lane < 40u) {
      win.left[lane] = s[min(i + lane, n - 1u)];
      win.right[lane] = s[j >= lane ? j - lane : 0u];
    }
    __builtin_amdgcn_wave_barrier();
    // (a, r) of this lane's probe, carried from batch to batch: row a holds len = lim + 1 - a
    uint32_t pa = 0, plen = lim + 1u, pr = lane;
    auto batch_term = [&](uint32_t m0) {
      float term = kNegInf;
      while (plen != 0u && pr >= plen) {
        pr -= plen;
        plen--;
        pa++;
      }
      const uint32_t a = pa, r = pr;
      pr += 64u;
      if (m0 + lane < nprobe) {
        const uint32_t k = i + 1u + a, l = j - 1u - r;
        const float x = qb[tri_off(n, l - k) + k];
        float sc;
        if (!CONTRA) {
          sc = turner_twoloop_flat(b.params->turner, a, r, win.left[0], win.right[0], win.left[1],
                                   win.left[2], win.right[1], win.right[2], win.left[1u + a],
                                   win.right[1u + r], win.right[r], win.left[a]);
        } else {
          // (same expression tree as Contra::twoloop, every table lookup issued at once)
          sc = contra_twoloop_flat(b.params->contra, a, r, win.left[0], win.right[0], win.left[1],
                                   win.right[1], win.left[1u + a], win.right[1u + r], win.right[r],
                                   win.left[a]);
        }
        if (x > kNegInf) term = x + sc;
      }
      return term;
    };
    float cur = batch_term(0u);
    for (uint32_t m0 = 0; m0 < nprobe; m0 += 64u) {
      const float nxt = batch_term(m0 + 64u);  // (all -inf past the last batch)
      const float t[1] = {pin(cur)};
      sum = fold_block<1>(sum, t, steps, P8);
      cur = nxt;
    }
  }
  const float mbc = model.mbclose(s, n, i, j);
  const float qm = (d >= 2) ? q.m[M_QM][tri_off(n, d - 2) + i + 1] : kNegInf;
  sum = lse_u(sum, qm + mbc, P8);
  const float acc = model.accessible(s, n, i, j);
  if (lane == 0u && sum > kNegInf) {
    q.m[M_MBC][o] = mbc;
    q.m[M_QB][o] = sum;
    q.m[M_QA][o] = sum + acc;
  }
}

// pair probability of one listed cell, first half: exterior term ⊕ enclosing 2-loops
// (src/mccaskill_algo.rs:559-593 / 663-700); the running sum is parked in the log-prob slot
// for outside_tail_lat.
template <bool CONTRA>
__device__ __forceinline__ void outside_head_lat(const DeviceBatch& b, const Seq& q, uint32_t d,
                                                 uint32_t i, const Piece8& P8) {
  const uint32_t n = q.n;
  const uint32_t j = i + d;
  const uint8_t* __restrict__ s = q.s;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t od = tri_off(n, d) + i;
  const float* __restrict__ qb = q.m[M_QB];
  const float* __restrict__ lp = q.m[M_P];
  const float qb_ij = qb[od];
  if (uni(__float_as_uint(qb_ij)) == 0xFF800000u) return;  // not in sums_close
  const float qa_ij = q.m[M_QA][od];
  const float* z = q.m[M_Z];
  const float ztot = z[tri_off(n, n - 1)];
  const float zl = (i < 1) ? 0.f : z[tri_off(n, i - 1)];                  // Z[0][i-1]
  const float zr = (j > n - 2) ? 0.f : z[tri_off(n, n - 2 - j) + j + 1];  // Z[j+1][n-1]
  float p = CONTRA ? zl + zr + qa_ij + b.params->contra.external_score_basepair - ztot
                   : zl + qa_ij + zr - ztot;
  __shared__ float steps[64];
  if (d + 2 < n) {
    // enclosing pairs (k,l) = (i-1-a, j+1+bb), a ascending (k descending), bb ascending,
    // a + bb <= 30, k >= 0, l <= n-1
    const uint32_t lim = min(static_cast<uint32_t>(RNAMC_MAX_2LOOP_LEN), n - 3 - d);
    // Turner: 64 probes per batch (its scorer costs ~1.8 us of dependent table lookups per
    // call however many lanes take part: 8 calls instead of 31).  CONTRAfold: row by row (its
    // scorer is cheap, and the shorter blocks are classified against a fresher sum):
    // measured either way, profiles/r02_latency_forms.txt
    if (!CONTRA) {
      // (64 probes per batch in fold order, as in inside_pair_lat; the bases around i
      // (leftwards) and j (rightwards) for turner_twoloop_flat)
      __shared__ Win2 win;
      if (lane < 40u) {
        win.left[lane] = s[i >= lane ? i - lane : 0u];
        win.right[lane] = s[min(j + lane, n - 1u)];
      }
      __builtin_amdgcn_wave_barrier();
      const uint32_t nprobe = (lim + 1u) * (lim + 2u) / 2u;
      uint32_t pa = 0, plen = lim + 1u, pr = lane;
      auto batch_term = [&](uint32_t m0) {
        float term = kNegInf;
        while (plen != 0u && pr >= plen) {
          pr -= plen;
          plen--;
          pa++;
        }
        const uint32_t a = pa, r = pr;
        pr += 64u;
        if (m0 + lane < nprobe) {
          if (a < i && j + 1u + r <= n - 1u) {
            const uint32_t k = i - 1u - a, l = j + 1u + r;
            const uint32_t x = tri_off(n, l - k) + k;
            const float qkl = qb[x];
            // (k,l) closes, (i,j) is enclosed
            const float sc = turner_twoloop_flat(
                b.params->turner, a, r, win.left[1u + a], win.right[1u + r], win.left[a],
                win.left[a >= 1u ? a - 1u : 0u], win.right[r], win.right[r >= 1u ? r - 1u : 0u],
                win.left[0], win.right[0], win.right[1], win.left[1]);
            if (qkl > kNegInf) term = lp[x] + qb_ij - qkl + sc;
          }
        }
        return term;
      };
      float cur = batch_term(0u);
      for (uint32_t m0 = 0; m0 < nprobe; m0 += 64u) {
        const float nxt = batch_term(m0 + 64u);
        const float t[1] = {pin(cur)};
        p = fold_block<1>(p, t, steps, P8);
        cur = nxt;
      }
    } else {
      // (the bases around i (leftwards) and j (rightwards) for contra_twoloop_flat: same
      // expression tree as Contra::twoloop, every table lookup issued at once)
      __shared__ Win2 winc;
      if (lane < 40u) {
        winc.left[lane] = s[i >= lane ? i - lane : 0u];
        winc.right[lane] = s[min(j + lane, n - 1u)];
      }
      __builtin_amdgcn_wave_barrier();
      auto row_term = [&](uint32_t a) {
        float term = kNegInf;
        if (a <= lim && a < i && lane <= lim - a && j + 1u + lane <= n - 1u) {
          const uint32_t k = i - 1u - a, l = j + 1u + lane;
          const uint32_t x = tri_off(n, l - k) + k;
          const float qkl = qb[x];
          // (k,l) closes, (i,j) is enclosed
          const float sc = contra_twoloop_flat(b.params->contra, a, lane, winc.left[1u + a], winc.right[1u + lane],
                                               winc.left[a], winc.right[lane], winc.left[0], winc.right[0],
                                               winc.right[1], winc.left[1]);
          if (qkl > kNegInf) term = lp[x] + qb_ij - qkl + sc;
        }
        return term;
      };
      float cur = row_term(0u);
      for (uint32_t a = 0; a <= lim; a++) {
        const float nxt = row_term(a + 1u);
        const float t[1] = {pin(cur)};
        p = fold_block<1>(p, t, steps, P8);
        cur = nxt;
      }
    }
  }
  if (lane == 0u) q.m[M_P][od] = p;
}

#endif  // RNAMC_LATENCY_H
