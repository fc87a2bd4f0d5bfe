// Throughput of one logsumexp fold step (src/utils.rs:579-627) on gfx950 when BOTH operands are
// known finite: forms with fewer / cheaper VALU instructions than the general lse() of
// rnamc_kernels.hip, each checked bit for bit against it on random finite operand pairs.
//  A : the general form (v_max, v_min, v_sub, 42-cell table + v_med3, cmp + cndmask + add, cubic,
//      final add, -inf fix-up cmp + cndmask)
//  F1: no fix-up; d = sum - x, z = |d| by AND, lo = v_min; small table + med3; cmp/cndmask/add
//  F2: F1 with a 2048-cell table indexed by the raw exponent bits (no clamp)
//  F3: F2 with the piece offset by arithmetic: t = thr' - z, boff = off + ((t >> 31) << 4)
//  F4: F3 with index (d >> 17) & 0x3FF8 and |d| as a source modifier (no AND, no shift-left)
//  F5: F4 but piece offset = off + ((t >> 27) & 16)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>

__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
constexpr float kNegInf = -__builtin_inff();
__constant__ float kCoef[9][4] = {
    {-0.0065591595f, 0.12764427f, 0.49965546f, 0.6931542f}, {-0.015515756f, 0.14467756f, 0.48829398f, 0.6958093f},
    {-0.012890925f, 0.13010283f, 0.51503986f, 0.6795586f}, {-0.0072142647f, 0.087754086f, 0.6208708f, 0.5909676f},
    {-0.0031455354f, 0.046722945f, 0.7592532f, 0.43487945f}, {-0.0010110698f, 0.018594341f, 0.88317305f, 0.25236955f},
    {-0.000196278f, 0.0046084408f, 0.9634432f, 0.09831489f}, {-0.0000113994f, 0.0003734731f, 0.9959107f, 0.0149855051f},
    {0.f, 0.f, 1.f, 0.f}};
__constant__ float kBreaks[8] = {0.66153675f, 1.6320158f, 2.4912589f, 3.3792500f, 4.426169f, 5.789071f, 7.8162727f, 11.862479f};

struct Tab {
  float4 coef[10];     // byte 0; row 9 = row 8 (identity) again: an overflowing d (= inf) lands there
  float2 cell[42];     // small table {thr, byte offset of the lower piece}
  float2 pad[5];
  float2 big[2048];    // {thr' = largest float below thr (FLT_MAX if none), byte offset of the lower piece}
  float2 bigc[2048];   // {thr, byte offset} for cmp forms
};


__device__ void load_tab(Tab* t) {
  for (unsigned x = threadIdx.x; x < 2048; x += blockDim.x) {
    if (x < 10) { const unsigned y = x < 9 ? x : 8; t->coef[x] = make_float4(kCoef[y][0], kCoef[y][1], kCoef[y][2], kCoef[y][3]); }
    if (x < 42) {
      float lo, hi;
      if (x == 0) { lo = 0; hi = 0.5f; } else if (x == 41) { lo = 16.f; hi = __builtin_inff(); }
      else { lo = __uint_as_float((0x3EFu + x) << 20); hi = __uint_as_float((0x3EFu + x + 1) << 20); }
      int piece = 0; for (int k = 0; k < 8; k++) piece += lo >= kBreaks[k];
      float thr = __builtin_inff(); if (piece < 8 && kBreaks[piece] < hi) thr = kBreaks[piece];
      t->cell[x] = make_float2(thr, __uint_as_float(piece * 16u));
    }
    {
      const float lo = __uint_as_float(x << 20);
      const float hi = (x + 1 < 2048) ? __uint_as_float((x + 1) << 20) : __builtin_inff();
      int piece = 0; for (int k = 0; k < 8; k++) piece += lo >= kBreaks[k];
      float thr = __builtin_inff();
      if (x < 0x7F8 && piece < 8 && kBreaks[piece] < hi) thr = kBreaks[piece];
      const float thrp = (thr == __builtin_inff()) ? 3.4028234664e38f : __uint_as_float(__float_as_uint(thr) - 1u);
      t->big[x] = make_float2(thrp, __uint_as_float(piece * 16u));
      t->bigc[x] = make_float2(thr, __uint_as_float(piece * 16u));
    }
  }
  __syncthreads();
}

__device__ __forceinline__ float poly(float4 co, float z) { return ((co.x * z + co.y) * z + co.z) * z + co.w; }

__device__ __forceinline__ float lseA(float sum, float x, const Tab* tab) {
  float hi = vmax(sum, x), lo = vmin(sum, x), z = hi - lo;
  int e = (int)(__float_as_uint(z) >> 20);
  int cell = min(max(e, 0x3EF), 0x418) - 0x3EF;
  float2 ce = tab->cell[cell];
  unsigned boff = __float_as_uint(ce.y) + ((z >= ce.x) ? 16u : 0u);
  float4 co = *(const float4*)((const char*)tab + boff);
  float r = lo + poly(co, z);
  return lo == kNegInf ? hi : r;
}
__device__ __forceinline__ float lseF1(float sum, float x, const Tab* tab) {
  const float d = sum - x;
  const float z = __uint_as_float(__float_as_uint(d) & 0x7FFFFFFFu);
  const float lo = vmin(sum, x);
  int e = (int)(__float_as_uint(z) >> 20);
  int cell = min(max(e, 0x3EF), 0x418) - 0x3EF;
  float2 ce = tab->cell[cell];
  unsigned boff = __float_as_uint(ce.y) + ((z >= ce.x) ? 16u : 0u);
  float4 co = *(const float4*)((const char*)tab + boff);
  return lo + poly(co, z);
}
__device__ __forceinline__ float lseF2(float sum, float x, const Tab* tab) {
  const float d = sum - x;
  const float z = __uint_as_float(__float_as_uint(d) & 0x7FFFFFFFu);
  const float lo = vmin(sum, x);
  const unsigned e8 = (__float_as_uint(z) >> 20) << 3;
  const float2 ce = *(const float2*)((const char*)tab->bigc + e8);
  unsigned boff = __float_as_uint(ce.y) + ((z >= ce.x) ? 16u : 0u);
  float4 co = *(const float4*)((const char*)tab + boff);
  return lo + poly(co, z);
}
__device__ __forceinline__ float lseF3(float sum, float x, const Tab* tab) {
  const float d = sum - x;
  const float z = __uint_as_float(__float_as_uint(d) & 0x7FFFFFFFu);
  const float lo = vmin(sum, x);
  const unsigned e8 = (__float_as_uint(z) >> 20) << 3;
  const float2 ce = *(const float2*)((const char*)tab->big + e8);
  const float t = ce.x - z;
  const unsigned boff = ((__float_as_uint(t) >> 31) << 4) + __float_as_uint(ce.y);
  float4 co = *(const float4*)((const char*)tab + boff);
  return lo + poly(co, z);
}
__device__ __forceinline__ float lseF4(float sum, float x, const Tab* tab) {
  const float d = sum - x;
  const float lo = vmin(sum, x);
  const unsigned e8 = (__float_as_uint(d) >> 17) & 0x3FF8u;
  const float2 ce = *(const float2*)((const char*)tab->big + e8);
  const float z = __builtin_fabsf(d);
  const float t = ce.x - z;
  const unsigned boff = ((__float_as_uint(t) >> 31) << 4) + __float_as_uint(ce.y);
  float4 co = *(const float4*)((const char*)tab + boff);
  return lo + poly(co, z);
}
__device__ __forceinline__ float lseF5(float sum, float x, const Tab* tab) {
  const float d = sum - x;
  const float lo = vmin(sum, x);
  const unsigned e8 = (__float_as_uint(d) >> 17) & 0x3FF8u;
  const float2 ce = *(const float2*)((const char*)tab->big + e8);
  const float z = __builtin_fabsf(d);
  const float t = ce.x - z;
  const unsigned boff = ((__float_as_uint(t) >> 27) & 16u) + __float_as_uint(ce.y);
  float4 co = *(const float4*)((const char*)tab + boff);
  return lo + poly(co, z);
}
// F6: F4 with cmp + cndmask + add for the piece offset (which of the two is cheaper)
__device__ __forceinline__ float lseF6(float sum, float x, const Tab* tab) {
  const float d = sum - x;
  const float lo = vmin(sum, x);
  const unsigned e8 = (__float_as_uint(d) >> 17) & 0x3FF8u;
  const float2 ce = *(const float2*)((const char*)tab->bigc + e8);
  const float z = __builtin_fabsf(d);
  unsigned boff = __float_as_uint(ce.y) + ((z >= ce.x) ? 16u : 0u);
  float4 co = *(const float4*)((const char*)tab + boff);
  return lo + poly(co, z);
}
// F7: F4 with lo from integer ops instead of v_min: lo = x + (d < 0 ? d : 0) is NOT exact; use
// the bit select  lo = d < 0 ? sum : x  as v_cndmask on the sign (cmp-free: v_ashrrev + v_bfi)
__device__ __forceinline__ float lseF7(float sum, float x, const Tab* tab) {
  const float d = sum - x;
  const unsigned m = (unsigned)((int)__float_as_uint(d) >> 31);  // all ones when sum < x
  const float lo = __uint_as_float((__float_as_uint(sum) & m) | (__float_as_uint(x) & ~m));
  const unsigned e8 = (__float_as_uint(d) >> 17) & 0x3FF8u;
  const float2 ce = *(const float2*)((const char*)tab->big + e8);
  const float z = __builtin_fabsf(d);
  const float t = ce.x - z;
  const unsigned boff = ((__float_as_uint(t) >> 31) << 4) + __float_as_uint(ce.y);
  float4 co = *(const float4*)((const char*)tab + boff);
  return lo + poly(co, z);
}

template <int V>
__device__ __forceinline__ float lseV(float s, float x, const Tab* tab) {
  if (V == 0) return lseA(s, x, tab);
  if (V == 1) return lseF1(s, x, tab);
  if (V == 2) return lseF2(s, x, tab);
  if (V == 3) return lseF3(s, x, tab);
  if (V == 4) return lseF4(s, x, tab);
  if (V == 5) return lseF5(s, x, tab);
  if (V == 6) return lseF6(s, x, tab);
  return lseF7(s, x, tab);
}

template <int V, int CH>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  __shared__ Tab tab;
  load_tab(&tab);
  float s[CH];
  for (int c = 0; c < CH; c++) s[c] = threadIdx.x * 0.01f + c;
  float x = 0.3f + threadIdx.x * 0.001f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        float xx = x + (float)(u + c) * 0.37f;
        s[c] = lseV<V>(s[c], xx, &tab);
      }
    }
    x += 0.001f;
  }
  float r = 0; for (int c = 0; c < CH; c++) r += s[c];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int V>
__global__ void __launch_bounds__(256) k_check(const float* a, const float* b, unsigned* bad, unsigned* first, int n) {
  __shared__ Tab tab;
  load_tab(&tab);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float r0 = lseA(a[i], b[i], &tab), r1 = lseV<V>(a[i], b[i], &tab);
    if (__float_as_uint(r0) != __float_as_uint(r1)) {
      if (atomicAdd(bad, 1u) == 0u) *first = (unsigned)i;
    }
  }
}

template <int V, int CH>
void run(const char* name) {
  float* out;
  hipMalloc(&out, 4 * 256 * 65536);
  const int iters = 400;
  printf("%-46s", name);
  for (int wps : {2, 4, 5, 6, 8}) {
    const int blocks = 256 * wps;  // 4 waves per block = one per SIMD
    hipLaunchKernelGGL((k<V, CH>), dim3(blocks), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, CH>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)iters * 8 * CH * wps);
    printf("  w%d %5.2f", wps, ns);
  }
  printf("   ns/lse/SIMD\n");
  hipFree(out);
}

template <int V>
void check(const char* name, const std::vector<float>& a, const std::vector<float>& b) {
  float *da, *db; unsigned *dbad, *dfirst;
  const int n = (int)a.size();
  hipMalloc(&da, 4 * n); hipMalloc(&db, 4 * n); hipMalloc(&dbad, 4); hipMalloc(&dfirst, 4);
  hipMemcpy(da, a.data(), 4 * n, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), 4 * n, hipMemcpyHostToDevice);
  hipMemset(dbad, 0, 4); hipMemset(dfirst, 0, 4);
  hipLaunchKernelGGL((k_check<V>), dim3(1024), dim3(256), 0, 0, da, db, dbad, dfirst, n);
  unsigned bad = 0, first = 0;
  hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
  hipMemcpy(&first, dfirst, 4, hipMemcpyDeviceToHost);
  printf("check %-8s vs A on %d finite pairs: %u differ", name, n, bad);
  if (bad) printf("  (first: sum=%.9g x=%.9g)", a[first], b[first]);
  printf("\n");
  hipFree(da); hipFree(db); hipFree(dbad); hipFree(dfirst);
}

int main() {
  // operand pairs: every exponent of z around the breakpoints, exact breakpoints, ties, zeros,
  // huge and tiny magnitudes
  std::vector<float> a, b;
  uint64_t st = 12345;
  auto rnd = [&]() { st += 0x9E3779B97F4A7C15ull; uint64_t z = st; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
  auto uni = [&]() { return (double)(rnd() >> 11) / 9007199254740992.0; };
  const float brk[8] = {0.66153675f, 1.6320158f, 2.4912589f, 3.3792500f, 4.426169f, 5.789071f, 7.8162727f, 11.862479f};
  for (int i = 0; i < 4000000; i++) {
    const int kind = i & 7;
    float s, x;
    if (kind < 4) {          // z uniform in [0, 14), base magnitude varied
      const float base = (float)((uni() - 0.5) * (kind == 0 ? 2.0 : kind == 1 ? 200.0 : kind == 2 ? 8000.0 : 2.0e6));
      const float z = (float)(uni() * 14.0);
      s = base; x = base - z;
      if (rnd() & 1) { float t = s; s = x; x = t; }
    } else if (kind == 4) {  // operands differing by exactly a breakpoint +- a few ulps
      const float t = brk[rnd() % 8];
      const int ul = (int)(rnd() % 5) - 2;
      const float z = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, t) + ul);
      s = (rnd() & 1) ? 0.f : z; x = (s == 0.f) ? -z : 0.f;
      if (rnd() & 1) { s = z; x = 0.f; }
    } else if (kind == 5) {  // wide z: 1e-30 .. 1e30
      const float z = (float)std::exp((uni() - 0.5) * 138.0);
      s = (float)(uni() * 10.0); x = s - z;
      if (rnd() & 1) { float t = s; s = x; x = t; }
    } else if (kind == 6) {  // equal operands, signed zeros, denormal differences
      s = (float)((uni() - 0.5) * 100.0); x = s;
      if ((rnd() & 3) == 0) { s = 0.f; x = -0.f; }
      if ((rnd() & 3) == 1) { x = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) + 1u); }
    } else {                 // huge magnitudes
      s = (float)((uni() - 0.5) * 6.0e38); x = (float)((uni() - 0.5) * 6.0e38);
      if (!std::isfinite(s - x)) x = s * 0.5f;
    }
    a.push_back(s); b.push_back(x);
  }
  check<1>("F1", a, b); check<2>("F2", a, b); check<3>("F3", a, b); check<4>("F4", a, b);
  check<5>("F5", a, b); check<6>("F6", a, b); check<7>("F7", a, b);
  run<0, 6>("A  general (6 chains)");
  run<1, 6>("F1 no fix-up, |d|, small table");
  run<2, 6>("F2 2048-cell table, no clamp");
  run<3, 6>("F3 arithmetic piece offset");
  run<4, 6>("F4 index by shift+mask, |d| modifier");
  run<5, 6>("F5 offset by (t>>27)&16");
  run<6, 6>("F6 F4 with cmp/cndmask offset");
  run<7, 6>("F7 F4, lo by bit select");
  run<0, 3>("A  general (3 chains)");
  run<4, 3>("F4 (3 chains)");
  return 0;
}
