// rnamc_probes.h — the 2-loop ("probe") fold of one cell, shared by the inside
// closing-pair block and the outside pair-probability block (device code,
// included by rnamc_kernels.hip only).
//
// Reference: the (k,l) double loops of src/mccaskill_algo.rs:306-325 / 412-436
// (inside: (i,j) closes, enclosed (k,l) = (i+1+a, j-1-b)) and 574-593 / 681-700
// (outside: (i,j) is enclosed, closing (k,l) = (i-1-a, j+1+b)); a ascending, then
// b ascending, a + b <= 30.  Scores: src/utils.rs:207-366 (Turner), 423-520
// (CONTRAfold).  The fold order and every f32 expression tree are the
// reference's; what is re-designed is how the operands reach the ALUs:
//  * bases come from two 32-base windows held in registers (2 bits per base);
//    the 4-bit slice {neighbour, base} of the varying pair indexes LDS directly;
//  * per class (1xmany / 2x3 / interior / bulge) one LDS table of float2
//    {terminal mismatch of the varying pair, its AU/GU end penalty} (Turner) or
//    {junction score of the varying pair, base-pair score} (CONTRAfold), laid out
//    [varying 4-bit slice along b][varying 4-bit slice along a];
//  * the length-dependent part (initiation + asymmetry, or the cumulative
//    length scores) is a 32x32 LDS table built once per block with the
//    reference's own operations;
//  * only the nine loops with a <= 2 and b <= 2 (stack, 1-bulges, 1x1, 1x2, 2x1,
//    2x2 tables) take the general, slower path;
//  * sums_close (and, outside, the log-probability) operands of the next 8
//    probes are in flight while the current 8 are folded.
#ifndef RNAMC_PROBES_H
#define RNAMC_PROBES_H

// (this header is included inside namespace rnamc { namespace { ... } })

struct ProbeTabs {
  float2 g[4][256];  // fused per-class tables, see header comment; g[3] = {0, penalty}
  float ii[32 * 32]; // length-dependent score by (a, b)
  float stack[256];
  float tmA[256];    // natural [p][q][u][v] layouts for the general path
  float tmB[256];
  float tmC[256];
  float misc[160];
};

__device__ __forceinline__ bool augu_code(int a, int b) {
  // AU UA GU UG as a 16-bit truth table over a*4+b
  return (0x5808u >> (a * 4 + b)) & 1u;
}

struct TurnerConsts {
  float pen, ninio_coeff, ninio_max;
};

// Turner: general path, any (a, b) — same tree as Turner::twoloop
__device__ __forceinline__ float turner_twoloop_general(const ProbeTabs& L,
                                                        const rnamc_turner_scores& t,
                                                        const TurnerConsts& k, const TwoLoopCodes& c,
                                                        uint32_t a, uint32_t b) {
  const float penc = augu_code(c.c0, c.c1) ? k.pen : 0.f;
  const float peni = augu_code(c.a0, c.a1) ? k.pen : 0.f;
  if (a == 0 && b == 0) return L.stack[idx4(c.c0, c.c1, c.a0, c.a1)];
  if (a == 0 || b == 0) {
    const uint32_t len = a + b;
    if (len == 1) return L.misc[1] + L.stack[idx4(c.c0, c.c1, c.a0, c.a1)];
    return L.misc[len] + penc + peni;
  }
  if (a == 1 && b == 1) return t.interior_scores_1x1[c.c0][c.c1][c.x1][c.y1][c.a0][c.a1];
  if (a == 1 && b == 2) return t.interior_scores_1x2[c.c0][c.c1][c.x1][c.y1][c.y2][c.a0][c.a1];
  if (a == 2 && b == 1) return t.interior_scores_1x2[c.a1][c.a0][c.y1][c.x2][c.x1][c.c1][c.c0];
  if (a == 2 && b == 2)
    return t.interior_scores_2x2[c.c0][c.c1][c.x1][c.y1][c.x2][c.y2][c.a0][c.a1];
  const uint32_t diff = a > b ? a - b : b - a;
  const int ic = idx4(c.c0, c.c1, c.x1, c.y1), ii = idx4(c.a1, c.a0, c.o1, c.o0);
  float mm;
  if (a == 1 || b == 1) {
    mm = L.tmA[ic] + L.tmA[ii];
  } else if ((a == 2 && b == 3) || (a == 3 && b == 2)) {
    mm = L.tmB[ic] + L.tmB[ii];
  } else {
    mm = L.tmC[ic] + L.tmC[ii];
  }
  return L.misc[32 + a + b] + fmaxf(k.ninio_coeff * static_cast<float>(diff), k.ninio_max) + mm +
         penc + peni;
}

// CONTRAfold: general path — same tree as Contra::twoloop
__device__ __forceinline__ float contra_twoloop_general(const ProbeTabs& L, const TwoLoopCodes& c,
                                                        uint32_t a, uint32_t b) {
  float sc;
  if (a == 0 && b == 0) {
    sc = L.stack[idx4(c.c0, c.c1, c.a0, c.a1)];
  } else {
    const float jsc = L.misc[c.c0 * 4 + c.c1] + L.tmA[idx4(c.c0, c.c1, c.x1, c.y1)];
    const float jsi = L.misc[c.a1 * 4 + c.a0] + L.tmA[idx4(c.a1, c.a0, c.o1, c.o0)];
    if (a == 0 || b == 0) {
      const uint32_t len = a + b;
      float s0 = 0.f;
      if (len == 1) s0 = L.tmB[16 + (a == 1 ? c.x1 : c.y1)];
      sc = s0 + L.misc[32 + len - 1] + jsc + jsi;
    } else {
      float s0;
      if (a == b) {
        const float s11 = (a + b == 2) ? L.tmB[c.x1 * 4 + c.y1] : 0.f;
        s0 = s11 + L.misc[96 + a - 1];
      } else {
        const uint32_t diff = a > b ? a - b : b - a;
        s0 = L.misc[112 + diff - 1];
      }
      const float se = (a <= RNAMC_MAX_INTERIOR_EXPLICIT && b <= RNAMC_MAX_INTERIOR_EXPLICIT)
                           ? L.misc[144 + (a - 1) * 4 + (b - 1)]
                           : 0.f;
      sc = s0 + se + L.misc[64 + a + b - 2] + jsc + jsi;
    }
  }
  return sc + L.misc[16 + c.a0 * 4 + c.a1];
}

// Fill the LDS tables.  OUTSIDE selects which pair the fused tables are keyed by:
//   inside  : varying = enclosed pair (k,l): t = s[l]*4 + s[l+1], r = s[k]*4 + s[k-1]
//   outside : varying = closing pair  (k,l): t = s[l]*4 + s[l-1], r = s[k+1]*4 + s[k]
template <bool CONTRA, bool OUTSIDE>
__device__ __forceinline__ void load_probe_tabs(ProbeTabs& L, const rnamc_params* P) {
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  if (!CONTRA) {
    const rnamc_turner_scores& T = P->turner;
    const float* tm[3] = {&T.terminal_mismatch_scores_1xmany[0][0][0][0],
                          &T.terminal_mismatch_scores_2x3[0][0][0][0],
                          &T.terminal_mismatch_scores_interior[0][0][0][0]};
    const float pen = T.helix_augu_end_penalty;
    for (uint32_t x = tid; x < 256; x += nt) {
      L.stack[x] = (&T.stack_scores[0][0][0][0])[x];
      L.tmA[x] = tm[0][x];
      L.tmB[x] = tm[1][x];
      L.tmC[x] = tm[2][x];
      const int t = x >> 4, r = x & 15;
      int p, q, u, v;  // TM[p][q][u][v], penalty of pair (pp, pq)
      int pp, pq;
      if (!OUTSIDE) {  // TM[al][ak][o1][o0], pen(ak, al)
        p = t >> 2, u = t & 3, q = r >> 2, v = r & 3;
        pp = q, pq = p;
      } else {         // TM[ck][cl][x1][y1], pen(ck, cl)
        q = t >> 2, v = t & 3, u = r >> 2, p = r & 3;
        pp = p, pq = q;
      }
      const int src = idx4(p, q, u, v);
      const float pe = augu_code(pp, pq) ? pen : 0.f;
      L.g[0][x] = make_float2(tm[0][src], pe);
      L.g[1][x] = make_float2(tm[1][src], pe);
      L.g[2][x] = make_float2(tm[2][src], pe);
      L.g[3][x] = make_float2(0.f, pe);
    }
    for (uint32_t x = tid; x < 31; x += nt) {
      L.misc[x] = T.bulge_scores_init[x];
      L.misc[32 + x] = T.interior_scores_init[x];
    }
    for (uint32_t x = tid; x < 1024; x += nt) {
      const uint32_t a = x >> 5, b = x & 31;
      float v = 0.f;
      if (a + b <= RNAMC_MAX_2LOOP_LEN) {
        if (a == 0 || b == 0) {
          v = T.bulge_scores_init[a + b];
        } else {
          const uint32_t diff = a > b ? a - b : b - a;
          v = T.interior_scores_init[a + b] +
              fmaxf(T.ninio_coeff * static_cast<float>(diff), T.ninio_max);
        }
      }
      L.ii[x] = v;
    }
  } else {
    const rnamc_fold_score_sets& F = P->contra;
    const float* tm = &F.terminal_mismatch_scores[0][0][0][0];
    const float* hc = &F.helix_close_scores[0][0];
    const float* bp = &F.basepair_scores[0][0];
    for (uint32_t x = tid; x < 256; x += nt) {
      L.stack[x] = (&F.stack_scores[0][0][0][0])[x];
      L.tmA[x] = tm[x];
      const int t = x >> 4, r = x & 15;
      if (!OUTSIDE) {  // junction_single((l,k)) = hc[al][ak] + tm[al][ak][o1][o0]; basepair[ak][al]
        const int al = t >> 2, o1 = t & 3, ak = r >> 2, o0 = r & 3;
        L.g[0][x] = make_float2(hc[al * 4 + ak] + tm[idx4(al, ak, o1, o0)], bp[ak * 4 + al]);
      } else {         // junction_single((k,l)) = hc[ck][cl] + tm[ck][cl][x1][y1]
        const int cl = t >> 2, y1 = t & 3, x1 = r >> 2, ck = r & 3;
        L.g[0][x] = make_float2(hc[ck * 4 + cl] + tm[idx4(ck, cl, x1, y1)], 0.f);
      }
    }
    for (uint32_t x = tid; x < 16; x += nt) {
      L.tmB[x] = (&F.interior_scores_1x1[0][0])[x];
      L.misc[x] = hc[x];
      L.misc[16 + x] = bp[x];
      L.misc[144 + x] = (&F.interior_scores_explicit[0][0])[x];
    }
    for (uint32_t x = tid; x < 4; x += nt) L.tmB[16 + x] = F.bulge_scores_0x1[x];
    for (uint32_t x = tid; x < RNAMC_MAX_LOOP_LEN; x += nt)
      L.misc[32 + x] = F.bulge_scores_len_cumulative[x];
    for (uint32_t x = tid; x < RNAMC_MAX_LOOP_LEN - 1; x += nt)
      L.misc[64 + x] = F.interior_scores_len_cumulative[x];
    for (uint32_t x = tid; x < RNAMC_MAX_INTERIOR_SYMMETRIC; x += nt)
      L.misc[96 + x] = F.interior_scores_symmetric_cumulative[x];
    for (uint32_t x = tid; x < RNAMC_MAX_INTERIOR_ASYMMETRIC; x += nt)
      L.misc[112 + x] = F.interior_scores_asymmetric_cumulative[x];
    for (uint32_t x = tid; x < 1024; x += nt) {
      const uint32_t a = x >> 5, b = x & 31;
      float v = 0.f;
      if (a + b <= RNAMC_MAX_LOOP_LEN && a + b >= 2) {
        if (a == 0 || b == 0) {
          // score(= 0.) + bulge_scores_len_cumulative[len - 1]
          v = 0.f + F.bulge_scores_len_cumulative[a + b - 1];
        } else if (!(a == 1 && b == 1)) {
          float s0;
          if (a == b) {
            s0 = 0.f + F.interior_scores_symmetric_cumulative[a - 1];
          } else {
            const uint32_t diff = a > b ? a - b : b - a;
            s0 = F.interior_scores_asymmetric_cumulative[diff - 1];
          }
          const float se = (a <= RNAMC_MAX_INTERIOR_EXPLICIT && b <= RNAMC_MAX_INTERIOR_EXPLICIT)
                               ? F.interior_scores_explicit[a - 1][b - 1]
                               : 0.f;
          v = s0 + se + F.interior_scores_len_cumulative[a + b - 2];
        }
      }
      L.ii[x] = v;
    }
  }
  __syncthreads();
}

// class of a non-special loop: 0 = 1xmany, 1 = 2x3, 2 = interior, 3 = bulge
__device__ __forceinline__ uint32_t loop_class(uint32_t a, uint32_t b) {
  if (a == 0 || b == 0) return 3u;
  if (a == 1 || b == 1) return 0u;
  if ((a == 2 && b == 3) || (a == 3 && b == 2)) return 1u;
  return 2u;
}

// pair-wise reversal of a 32-base window: position q of the result holds base 31-q
__device__ __forceinline__ uint64_t reverse_pairs(uint64_t w) {
  uint64_t x = __builtin_bitreverse64(w);
  return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}

// Per-lane state that is fixed over the fold of one cell.
struct ProbeFixed {
  float tm0, tm1, tm2;  // Turner: terminal mismatch of the fixed pair for classes 0..2
  float pen;            // Turner: AU/GU penalty of the fixed pair
  float js;             // CONTRAfold: junction_single of the fixed pair
  float bp;             // CONTRAfold outside: basepair score of the (fixed) enclosed pair
};

// The fold.  Returns the updated running sum.
//   OUTSIDE = false: sum ⊕= sums_close(k,l) + twoloop(i,j,k,l)
//   OUTSIDE = true : sum ⊕= ((logp(k,l) + qb_ij) - sums_close(k,l)) + twoloop(k,l,i,j)
// `act`: this lane owns a cell that takes part; lim: largest a+b with a probe on
// this diagonal (uniform); for OUTSIDE each lane additionally needs k >= 0 and
// l <= n-1.
template <bool CONTRA, bool OUTSIDE>
__device__ __forceinline__ float probe_fold(const DeviceBatch& b, const Seq& q, uint32_t d,
                                            uint32_t i, bool act, uint32_t lim, float sum,
                                            float qb_ij, const LseTab* tab, const ProbeTabs& L) {
  const uint32_t n = q.n;
  const uint32_t j = i + d;
  const float* __restrict__ qb = q.m[M_QB];
  const float2* __restrict__ pq = reinterpret_cast<const float2*>(q.m[M_PQ]);
  const uint32_t i4 = i * 4u;
  // windows: wa walks with a (the k side), wb walks with b (the l side)
  //  inside : wa = bases i .. i+31 (k-1 at position a, k at a+1)
  //           wb = bases j .. j-31 reversed (l+1 at position b, l at b+1)
  //  outside: wa = bases i-31 .. i (k at position 30-a, k+1 at 31-a)
  //           wb = bases j .. j+31 (l-1 at position b, l at b+1)
  const Win wl = load_win(q.pk, OUTSIDE ? static_cast<int>(i) - 31 : static_cast<int>(i));
  const Win wr = load_win(q.pk, OUTSIDE ? static_cast<int>(j) : static_cast<int>(j) - 31);
  const uint64_t wa64 = (static_cast<uint64_t>(wl.hi) << 32) | wl.lo;
  const uint64_t wr64 = (static_cast<uint64_t>(wr.hi) << 32) | wr.lo;
  const uint64_t wb64 = OUTSIDE ? wr64 : reverse_pairs(wr64);

  TwoLoopCodes c;
  ProbeFixed fx;
  TurnerConsts tk{0.f, 0.f, 0.f};
  if (!CONTRA) {
    tk.pen = b.params->turner.helix_augu_end_penalty;
    tk.ninio_coeff = b.params->turner.ninio_coeff;
    tk.ninio_max = b.params->turner.ninio_max;
  }
  if (!OUTSIDE) {
    c.c0 = wbase(wl, 0);
    c.c1 = wbase(wr, 31);
    c.x1 = wbase(wl, 1);
    c.y1 = wbase(wr, 30);
    c.x2 = wbase(wl, 2);
    c.y2 = wbase(wr, 29);
    const int ic = idx4(c.c0, c.c1, c.x1, c.y1);
    if (!CONTRA) {
      fx.tm0 = L.tmA[ic];
      fx.tm1 = L.tmB[ic];
      fx.tm2 = L.tmC[ic];
      fx.pen = augu_code(c.c0, c.c1) ? tk.pen : 0.f;
    } else {
      fx.js = L.misc[c.c0 * 4 + c.c1] + L.tmA[ic];
    }
  } else {
    c.a0 = wbase(wl, 31);
    c.a1 = wbase(wr, 0);
    c.o0 = wbase(wl, 30);
    c.o1 = wbase(wr, 1);
    const int iin = idx4(c.a1, c.a0, c.o1, c.o0);
    if (!CONTRA) {
      fx.tm0 = L.tmA[iin];
      fx.tm1 = L.tmB[iin];
      fx.tm2 = L.tmC[iin];
      fx.pen = augu_code(c.a0, c.a1) ? tk.pen : 0.f;
    } else {
      fx.js = L.misc[c.a1 * 4 + c.a0] + L.tmA[iin];
      fx.bp = L.misc[16 + c.a0 * 4 + c.a1];
    }
  }

  // operand addresses: pair (k,l)
  //  inside : diagonal d-2-a-bb, offset i+1+a      outside: diagonal d+2+a+bb, offset i-1-a
  // Loads are unconditional: a probe past the row's end re-reads the row's last
  // probe (uniform clamp), a lane whose (k,l) falls off the sequence reads a
  // neighbouring diagonal (in bounds: 64-float pad) and is masked by value below.
  auto ubase = [&](const float* m, uint32_t a, uint32_t bb) {
    return OUTSIDE ? m + tri_off(n, d + 2 + a + bb) - 1 - a : m + tri_off(n, d - 2 - a - bb) + 1 + a;
  };
  // chunk iterator over rows: (row, first b); rows are cut into chunks of kPU
  auto chunk_next = [&](uint32_t& ra, uint32_t& rb) {
    rb += kPU;
    if (ra + rb > lim) {
      ra++;
      rb = 0;
    }
  };
  struct PBuf {
    float xs[kPU], ps[kPU];
  };
  auto fetch = [&](PBuf& B, uint32_t ra, uint32_t rb) {
#ifdef RNAMC_PROBE_RESIDENT
    // timing experiment (results wrong by construction): every probe of the cell re-reads the
    // cell's first probe, i.e. the 2-loop blocks draw next to nothing from HBM — the upper bound of
    // what ANY scheme that keeps their operands on chip or compacted could gain
    ra = 0;
    rb = 0;
    const uint32_t blast = 0;
#else
    const uint32_t blast = lim - ra;  // last probe of the row
#endif
#pragma unroll
    for (int u = 0; u < kPU; u++) {
      const uint32_t bb = min(rb + static_cast<uint32_t>(u), blast);
      if (OUTSIDE) {
        // {log prob, sums_close} of the enclosing pair with ONE 8-byte gather (M_PQ)
        const float2* base2 = pq + tri_off(n, d + 2 + ra + bb) - 1 - ra;
        const float2 v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base2) + 2u * i4);
        B.ps[u] = v.x;
        B.xs[u] = v.y;
      } else {
        B.xs[u] = ldu(ubase(qb, ra, bb), i4);
        B.ps[u] = 0.f;
      }
    }
  };
  // outside: lane-level validity of (k,l): k = i-1-a >= 0 and l = j+1+bb <= n-1
  const uint32_t bmax = (OUTSIDE && act && j + 2 <= n) ? n - 2 - j : 0u;  // largest valid bb
  const bool lane_l_ok = act && (!OUTSIDE || j + 2 <= n);
  const char* gbase = reinterpret_cast<const char*>(&L.g[0][0]);

  // one chunk: probes (a, b0 .. b0+kPU-1).  KIND (compile time) keeps the per-probe
  // conditions out of the common case: 0 = first chunk of a row (the small loops and the
  // loop classes of b <= 3 live here), 1 = a later, full chunk (every probe is a generic
  // interior loop or the row's bulge / 1xn tail class), 2 = a later, partial chunk,
  // 3 = decide at run time.
  auto do_chunk = [&](const PBuf& B, uint32_t a, uint32_t b0, auto kind) {
    constexpr int KIND = decltype(kind)::value;
    // row constants: 4-bit slice of the a-side window -> byte offset of r in a float2 row
    const uint32_t rs = OUTSIDE ? static_cast<uint32_t>(wa64 >> (2u * (30u - a))) & 15u
                                : static_cast<uint32_t>(wa64 >> (2u * a)) & 15u;
    const uint32_t r8 = rs << 3;
    if (!OUTSIDE) {
      c.a0 = static_cast<int>(rs >> 2);
      c.o0 = static_cast<int>(rs & 3u);
    } else {
      c.c0 = static_cast<int>(rs & 3u);
      c.x1 = static_cast<int>(rs >> 2);
      c.x2 = wbase(wl, (32u - a) & 31u);  // s[k+2], read only when a >= 1
    }
    // class of the row's long tail (b >= 4)
    const uint32_t rcls = (a == 0) ? 3u : (a == 1 ? 0u : 2u);
    const float rtm = (a == 0) ? 0.f : (a == 1 ? fx.tm0 : fx.tm2);
    uint64_t wcur = wb64 >> (2u * b0);
    const uint32_t rowlen = lim - a + 1;
    const bool row_ok = lane_l_ok && (!OUTSIDE || a < i);
#pragma unroll
    for (int u = 0; u < kPU; u++) {
      const uint32_t bb = b0 + u;
      const uint32_t ts = static_cast<uint32_t>(wcur) & 15u;
      wcur >>= 2;
      if (KIND == 1 || bb < rowlen) {
        float y;
        if ((KIND == 0 || (KIND == 3 && b0 == 0)) && u <= 2 && a <= 2) {
          // the nine small loops: general path
          if (!OUTSIDE) {
            c.a1 = static_cast<int>(ts >> 2);
            c.o1 = static_cast<int>(ts & 3u);
          } else {
            c.c1 = static_cast<int>(ts >> 2);
            c.y1 = static_cast<int>(ts & 3u);
            c.y2 = wbase(wr, (bb + 31u) & 31u);  // s[l-2], read only when bb >= 1
          }
          y = CONTRA ? contra_twoloop_general(L, c, a, bb)
                     : turner_twoloop_general(L, b.params->turner, tk, c, a, bb);
        } else {
          uint32_t cls = rcls;
          float ftm = rtm;
          if ((KIND == 0 || (KIND == 3 && b0 == 0)) && u <= 3 && !CONTRA) {
            cls = loop_class(a, bb);
            ftm = (cls == 3u) ? 0.f : (cls == 0u ? fx.tm0 : (cls == 1u ? fx.tm1 : fx.tm2));
          }
          const float2 gv = *reinterpret_cast<const float2*>(
              gbase + (CONTRA ? 0u : cls * 2048u) + (ts << 7) + r8);
          const float iiv = L.ii[a * 32u + bb];
          if (!CONTRA) {
            // INIT[len] + ninio + (TM(close) + TM(enclosed)) + pen(close) + pen(enclosed)
            const float mm = OUTSIDE ? gv.x + ftm : ftm + gv.x;
            y = OUTSIDE ? ((iiv + mm) + gv.y) + fx.pen : ((iiv + mm) + fx.pen) + gv.y;
          } else {
            // (len part + junction_single(close)) + junction_single(enclosed) + basepair
            y = OUTSIDE ? ((iiv + gv.x) + fx.js) + fx.bp : ((iiv + fx.js) + gv.x) + gv.y;
          }
        }
        if (!OUTSIDE) {
          sum = lse(sum, B.xs[u] + y, tab);
        } else {
          // absent pair (sums_close = -inf) or (k,l) off the sequence: no term
          const float term = B.ps[u] + qb_ij - B.xs[u] + y;
          const bool hit = row_ok && bb <= bmax && B.xs[u] > kNegInf;
          sum = lse(sum, hit ? term : kNegInf, tab);
        }
      }
    }
  };

  // two-stage pipeline over the chunk sequence (rows a ascending, chunks b0 ascending).
  // Inside, the next chunk is fetched unconditionally (past the end: the last row's first
  // probe again), so the number of loads in flight at every wait is static; outside (two
  // operands per probe) the conditional form measured faster.
  auto fold_chunk = [&](const PBuf& B, uint32_t a, uint32_t b0) {
    if (OUTSIDE) {  // one body with run-time conditions (3): measured faster there
      do_chunk(B, a, b0, std::integral_constant<int, 3>{});
    } else if (b0 == 0) {
      do_chunk(B, a, b0, std::integral_constant<int, 0>{});
    } else if (b0 + kPU <= lim - a + 1) {
      do_chunk(B, a, b0, std::integral_constant<int, 1>{});
    } else {
      do_chunk(B, a, b0, std::integral_constant<int, 2>{});
    }
  };
  PBuf A, B;
  uint32_t a = 0, b0 = 0;    // chunk being folded
  uint32_t fa = 0, fb = 0;   // next chunk to fetch
  fetch(A, fa, fb);
  chunk_next(fa, fb);
  for (;;) {
    bool more = fa <= lim;
    if (!OUTSIDE || more) fetch(B, more ? fa : lim, more ? fb : 0u);
    fold_chunk(A, a, b0);
    if (!more) break;
    a = fa;
    b0 = fb;
    chunk_next(fa, fb);
    more = fa <= lim;
    if (!OUTSIDE || more) fetch(A, more ? fa : lim, more ? fb : 0u);
    fold_chunk(B, a, b0);
    if (!more) break;
    a = fa;
    b0 = fb;
    chunk_next(fa, fb);
  }
  return sum;
}

#endif  // RNAMC_PROBES_H
