// rnamc_tree_mx.h — the banded mid-field of the cubic products on the matrix cores (included by
// rnamc_tree.hip; same contract as k_tree_mid: for every cell of a band of diagonals the {max, sum}
// pair of the terms that were final before the band started, into the mid-field ring).
//
// A (logsumexp, +) product IS a matrix product under exponentiation:
//   (+)_k A(i,k) + B(k,j)  =  S + ln  SUM_k  exp(A(i,k) - sa(i)) * exp(B(k,j) - sb(j)),   S = sa(i) + sb(j)
// k_tree_mid pays ~10 VALU slots a term for it (a running maximum and an exp2 per TERM: half of the
// slots are the quarter-rate transcendentals).  Here the exponentials are taken per OPERAND ELEMENT
// — once for the 32 cells of a tile row or column that use it — and a term is one multiply-add of
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: exact f32 products, a k-ordered fmaf chain, denormals
// kept; MI355X guide, "FP32-input MFMA").  What makes that safe in f32 is the scale: not one per
// matrix (ln-values span thousands of nats over a row) but one per operand row and CHUNK of 32 k,
// an integer power of two 2^E with the chunk's largest factor in (2^47, 2^48]: factors of a chunk
// keep full precision down to 2^-126, i.e. while they lie within 120 nats of the chunk's maximum
// (sums_1ormore_basepairs grows by at most a stacked pair's ~6 nats per base), products reach 2^96
// at most, and a chunk's partial sum joins the cell's running (exponent, sum) pair by two v_ldexp
// — no transcendental per cell either.
//
// Shape: a workgroup owns one 32 x 32 tile of cells in (i, j) — rows i0 .. i0+31, columns
// j0 = i0 + dlo + 32 c .. — of the band's parallelogram (32 rows need 31 + band width columns: three
// tiles for 64 diagonals, two thirds of whose cells are the band's), its four waves take the chunks of
// the tile's k range in turn and meet in LDS at the end, in a fixed order (deterministic).  Lane
// (r, h) holds row r of A and column r of B at the 16 consecutive k = 32 q + 16 h + e of chunk q —
// exactly the MFMA's operand map (lane l: A[l & 31][l >> 5], B[l >> 5][l & 31]) when step e of a
// chunk multiplies the k pair {32 q + e, 32 q + 16 + e}: operands go from global memory through the
// lane's own registers into the matrix core, no LDS staging, no cross-lane traffic but the row maximum
// (one exchange between the halves).  The three products differ in where an operand row starts and
// in its valid k interval (an element outside it is exp(-inf) = 0: the masks are separable, so the
// inner loop knows nothing of them):
//   prod 0 (inside)  C(i,j) = (+)_{k = j-T+1 .. i+T}  Q1(i,k-1) + Zr_mb(k,j)
//   prod 1 (outside) C(i,j) = (+)_{k = i+T .. n-1}    W(i,k)    + Q1(j+1,k-1)
//   prod 2 (outside) C(i,j) = (+)_{k = 0 .. j-T}      Q1(k+1,i-1) + R(k,j)
// (src/mccaskill_algo.rs:344-351, 540-543, 594-601 with both operands' spans restricted as k_tree_mid
// documents.)

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // (rows start at any k: dword-aligned x4 loads)
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kMxWaves = 4;   // k slices of one tile
constexpr int kMxBias = 48;   // a chunk's largest factor lies in (2^47, 2^48]
constexpr int kMxNone = -(1 << 28);  // exponent of an operand row without a finite element / of an empty sum
constexpr float kL2Ehi = 1.4426950216293335f;  // float(log2 e)
constexpr float kL2Elo = 1.9259629911e-8f;     // log2 e - kL2Ehi
constexpr float kLn2hi = 0.693359375f;         // 355 / 512: integer * kLn2hi is exact below 2^15
constexpr float kLn2lo = -2.12194440e-4f;      // ln 2 - kLn2hi

// 16 consecutive elements of an operand row at k = kb .. kb + 15 (p points at k = 0 of the row), the
// ones outside [lo, hi] replaced by -inf when MASKED
template <bool MASKED>
__device__ __forceinline__ void mx_load(float (&v)[16], const float* __restrict__ p, int kb, int lo, int hi) {
#pragma unroll
  for (int x = 0; x < 4; x++) {
    const f32x4u t = *reinterpret_cast<const f32x4u*>(p + kb + 4 * x);
    v[4 * x] = t.x;
    v[4 * x + 1] = t.y;
    v[4 * x + 2] = t.z;
    v[4 * x + 3] = t.w;
  }
  if (MASKED) {
    const uint32_t rel = static_cast<uint32_t>(kb - lo), len = static_cast<uint32_t>(hi - lo);  // (lo <= hi + 1)
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = (hi >= lo && rel + static_cast<uint32_t>(e) <= len) ? v[e] : kNegInf;
  }
}

// the row's factors 2^(v log2 e - E) in place; E (both halves of the wave agree on it)
__device__ __forceinline__ int mx_scale(float (&v)[16], uint32_t lane) {
  float mu = __builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), v[2]);
#pragma unroll
  for (int e = 3; e < 15; e += 2) mu = __builtin_fmaxf(__builtin_fmaxf(mu, v[e]), v[e + 1]);
  mu = __builtin_fmaxf(mu, v[15]);
  mu = __builtin_fmaxf(mu, __int_as_float(__builtin_amdgcn_ds_bpermute(static_cast<int>((lane ^ 32u) << 2),
                                                                       __float_as_int(mu))));
  const bool has = mu > kNegInf;
  const float ms = has ? mu : 0.f;
  const float lo = ms * kL2Elo;
  const float ef = __builtin_ceilf(__builtin_fmaf(ms, kL2Ehi, lo));
  // c = (mu log2 e - ceil(mu log2 e)) + bias, the first difference in one rounding
  const float c = (__builtin_fmaf(ms, kL2Ehi, -ef) + lo) + static_cast<float>(kMxBias);
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = ex2(__builtin_fmaf(v[e] - ms, kL2Ehi, c));
  return has ? static_cast<int>(ef) - kMxBias : kMxNone;
}

// KS = 4: the workgroup's four waves share one tile (k slices, merged through LDS) — few tiles, long k ranges:
// a lone sequence; KS = 1: a tile per wave, its whole k range in one software pipeline — a batch's thousands
// of tiles, whose k ranges are a few chunks (one wave's first loads are the exposed part of a tile's time).
template <int KS>
__global__ void __launch_bounds__(64 * kMxWaves)
k_tree_mid_mx(TreeBatch b, uint32_t dlo, uint32_t dhi, uint32_t thr, int outside, uint32_t nbi, uint32_t ntc) {
  __shared__ int ea_lds[kMxWaves][32];
  __shared__ float2 red[KS == 1 ? 1 : kMxWaves][16][64];  // {exponent (as int bits), sum}
  const TSeq q = load_tseq(b, blockIdx.y);
  const int n = static_cast<int>(q.n);
  const uint32_t ld = q.ld;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)));
  // (Workgroups go to the 8 XCDs round-robin by id.  Dealing them so that tile neighbours — which share B
  // columns — meet in one L2 was measured twice, whole eighths of the tile order per XCD and runs of four
  // workgroups: 10-20 % SLOWER passes both times, profiles/r04_tree_batch_route.txt; the plain order stays.)
  const uint32_t bx = blockIdx.x;
  const uint32_t tile = KS == 1 ? bx * kMxWaves + wave : bx;
  if (KS != 1 && tile >= nbi * ntc * (outside ? 2u : 1u)) return;  // (padding; uniform)
  const uint32_t tc = tile % ntc, rest = tile / ntc;
  const uint32_t bi = rest % nbi;
  const int prod = outside ? 1 + static_cast<int>(rest / nbi) : 0;
  if (KS == 1 && rest / nbi >= (outside ? 2u : 1u)) return;  // (the last workgroup's spare waves)
  const int i0 = 32 * static_cast<int>(bi), j0 = i0 + static_cast<int>(dlo) + 32 * static_cast<int>(tc);
  const int dtop = min(static_cast<int>(dhi), n - 1);
  // (uniform in the wave, and in the workgroup when it shares the tile: no cell of the band in this tile)
  if (j0 >= n || j0 - (i0 + 31) > dtop) return;
  const int T = static_cast<int>(thr);
  const int r = static_cast<int>(lane & 31u), h = static_cast<int>(lane >> 5);
  const int i = i0 + r, j = j0 + r;

  // the lane's operand rows (k = 0 of each) and their valid k intervals; the tile's k range
  const float* pa;
  const float* pb;
  int alo, ahi, blo, bhi, klo, khi;
  if (prod == 0) {
    const bool va = i < n, vb = j < n;
    pa = q.m[T_Q1R] + static_cast<size_t>(va ? i : 0) * ld - 1;
    pb = q.m[T_ZRM] + static_cast<size_t>(vb ? j : 0) * ld;
    alo = va ? i + 1 : 1;
    ahi = va ? i + T : 0;
    blo = vb ? max(j - T + 1, 1) : 1;
    bhi = vb ? j : 0;
    klo = max(j0 - T + 1, max(i0 + 1, 1));
    khi = min(min(i0 + 31 + T, j0 + 31), n - 1);
  } else if (prod == 1) {
    const bool va = i < n, vb = j + 1 < n;
    pa = q.m[T_ZRE] + static_cast<size_t>(va ? i : 0) * ld;
    pb = q.m[T_Q1R] + static_cast<size_t>(vb ? j + 1 : 0) * ld - 1;
    alo = va ? i + T : 1;
    ahi = va ? n - 1 : 0;
    blo = vb ? j + 2 : 1;
    bhi = vb ? n - 1 : 0;
    klo = max(i0 + T, j0 + 2);
    khi = n - 1;
  } else {
    const bool va = i >= 1 && i < n, vb = j < n;
    pa = q.m[T_Q1C] + static_cast<size_t>(va ? i - 1 : 0) * ld + 1;
    pb = q.m[T_ZRM] + static_cast<size_t>(vb ? j : 0) * ld;
    alo = va ? 0 : 1;
    ahi = va ? i - 2 : 0;
    blo = vb ? 0 : 1;
    bhi = vb ? j - T : 0;
    klo = 0;
    khi = min(min(i0 + 29, j0 + 31 - T), n - 1);
  }
  if (ahi < alo) { alo = 1; ahi = 0; }
  if (bhi < blo) { blo = 1; bhi = 0; }

  // the cells this lane's accumulators stand for: column j0 + r, rows i0 + 8 g + 4 h + t (reg = 4 g + t)
  int em[16];
  float sm[16];
#pragma unroll
  for (int x = 0; x < 16; x++) {
    em[x] = kMxNone;
    sm[x] = 0.f;
  }
  if (klo <= khi) {
    const int qlo = klo >> 5, qhi = khi >> 5;
    float ra[16], rb[16];
    bool inner = false;
    auto fetch = [&](int qq) {
      const int kb = 32 * qq + 16 * h;
      // (uniform: every lane's intervals hold the whole chunk)
      inner = __builtin_amdgcn_ballot_w64(alo <= 32 * qq && 32 * qq + 31 <= ahi && blo <= 32 * qq && 32 * qq + 31 <= bhi) ==
              ~0ull;
      if (inner) {
        mx_load<false>(ra, pa, kb, alo, ahi);
        mx_load<false>(rb, pb, kb, blo, bhi);
      } else {
        mx_load<true>(ra, pa, kb, alo, ahi);
        mx_load<true>(rb, pb, kb, blo, bhi);
      }
    };
    int qq = qlo + (KS == 1 ? 0 : static_cast<int>(wave));
    if (qq <= qhi) fetch(qq);
    for (; qq <= qhi; qq += KS) {
      const int ea = mx_scale(ra, lane), eb = mx_scale(rb, lane);
      float fa[16], fb[16];
#pragma unroll
      for (int e = 0; e < 16; e++) {
        fa[e] = ra[e];
        fb[e] = rb[e];
      }
      if (qq + KS <= qhi) fetch(qq + KS);
      f32x16 c;
#pragma unroll
      for (int x = 0; x < 16; x++) c[x] = 0.f;
#pragma unroll
      for (int e = 0; e < 16; e++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], c, 0, 0, 0);
      // row exponents of the accumulator's 16 rows: through the wave's own 128 bytes of LDS
      __builtin_amdgcn_wave_barrier();
      if (h == 0) ea_lds[wave][r] = ea;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const i32x4 e4 = *reinterpret_cast<const i32x4*>(&ea_lds[wave][8 * g + 4 * h]);
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int x = 4 * g + t;
          const int M = e4[t] + eb;
          const int mn = max(em[x], M);
          sm[x] = __builtin_ldexpf(sm[x], em[x] - mn) + __builtin_ldexpf(c[x], M - mn);
          em[x] = mn;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  auto finish = [&](int x, int mn, float s) {  // register x of the tile: row 8 (x / 4) + 4 h + x % 4, column r
    const int ci = i0 + 8 * (x >> 2) + 4 * h + (x & 3), cj = j0 + r;
    const int d = cj - ci;
    if (ci < n && cj < n && d >= static_cast<int>(dlo) && d <= dtop) {
      float2 o = make_float2(kEmpty, 0.f);
      if (s > 0.f) {
        // s 2^mn = (2 f) 2^(mn + e - 1), f in [0.5, 1); the exponent goes to nats in two exact pieces
        const int I = mn + __builtin_amdgcn_frexp_expf(s) - 1;
        const float fi = static_cast<float>(I);
        o.x = fi * kLn2hi;
        o.y = 2.f * __builtin_amdgcn_frexp_mantf(s) * ex2(fi * (kLn2lo * kL2E));
      }
      q.mid[(static_cast<size_t>(prod) * b.ring + static_cast<uint32_t>(d) % b.ring) * q.vec + static_cast<uint32_t>(ci)] = o;
    }
  };
  if (KS == 1) {
#pragma unroll
    for (int x = 0; x < 16; x++) finish(x, em[x], sm[x]);
    return;
  }
  // the four waves' partial sums meet: wave w finishes registers 4 w .. 4 w + 3 (rows 8 w + 4 h + t)
#pragma unroll
  for (int x = 0; x < 16; x++) red[wave][x][lane] = make_float2(__int_as_float(em[x]), sm[x]);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int x = 4 * static_cast<int>(wave) + t;
    int mn = kMxNone;
    float2 v[kMxWaves];
#pragma unroll
    for (int w = 0; w < kMxWaves; w++) {
      v[w] = red[w][x][lane];
      mn = max(mn, __float_as_int(v[w].x));
    }
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kMxWaves; w++) s += __builtin_ldexpf(v[w].y, __float_as_int(v[w].x) - mn);
    finish(x, mn, s);
  }
}

}  // namespace

void launch_tree_mid_mx(const TreeBatch& b, bool outside, uint32_t dlo, uint32_t dhi, uint32_t thr, uint32_t max_n,
                        uint32_t nseq, hipStream_t st) {
  if (dlo >= max_n || dhi < dlo || nseq == 0) return;
  const uint32_t nbi = (max_n - dlo + 31u) / 32u;
  const uint32_t ntc = (31u + (dhi - dlo + 1u) + 31u) / 32u;
  const uint32_t tiles = nbi * ntc * (outside ? 2u : 1u);
  if (static_cast<uint64_t>(tiles) * nseq >= 4096u)
    hipLaunchKernelGGL(k_tree_mid_mx<1>, dim3((tiles + kMxWaves - 1) / kMxWaves, nseq, 1), dim3(64 * kMxWaves),
                       0, st, b, dlo, dhi, thr, outside ? 1 : 0, nbi, ntc);
  else
    hipLaunchKernelGGL(k_tree_mid_mx<kMxWaves>, dim3(tiles, nseq, 1), dim3(64 * kMxWaves), 0, st, b, dlo, dhi,
                       thr, outside ? 1 : 0, nbi, ntc);
}
