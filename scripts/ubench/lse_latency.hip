// Dependent-chain latency of one logsumexp fold step on gfx950, three variants:
//  A: cell-LUT piece lookup (2 dependent ds_reads)     B: compare ladder + 1 ds_read
//  C: all-VALU select tree (no LDS)
// One wave per block; occupancy varied by grid size (blocks per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

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
struct Tab { float4 coef[9]; float2 cell[42]; float4 fused[42][3]; };
__device__ void load_tab(Tab* t) {
  unsigned x = threadIdx.x;
  if (x < 9) t->coef[x] = make_float4(kCoef[x][0], kCoef[x][1], kCoef[x][2], kCoef[x][3]);
  if (x < 42) {
    float lo, hi;
    if (x == 0) { lo = 0; hi = 0.5f; } else if (x == 41) { lo = 16.f; hi = __builtin_inff(); }
    else { lo = __uint_as_float((0x3EFu + x) << 20); hi = __uint_as_float((0x3EFu + x + 1) << 20); }
    int piece = 0; for (int k = 0; k < 8; k++) piece += lo >= kBreaks[k];
    float thr = __builtin_inff(); if (piece < 8 && kBreaks[piece] < hi) thr = kBreaks[piece];
    t->cell[x] = make_float2(thr, __uint_as_float(piece * 16u));
    int hi_piece = piece < 8 ? piece + 1 : 8;
    t->fused[x][0] = make_float4(thr, 0.f, 0.f, 0.f);
    t->fused[x][1] = make_float4(kCoef[piece][0], kCoef[piece][1], kCoef[piece][2], kCoef[piece][3]);
    t->fused[x][2] = make_float4(kCoef[hi_piece][0], kCoef[hi_piece][1], kCoef[hi_piece][2], kCoef[hi_piece][3]);
  }
  __syncthreads();
}
__device__ __forceinline__ float lseA(float sum, float x, const Tab* tab) {
  float hi = vmax(sum, x), lo = vmin(sum, x), z = hi - lo;
  int e = (int)(__float_as_uint(z) >> 20);
  int cell = min(max(e, 0x3EF), 0x418) - 0x3EF;
  float2 ce = tab->cell[cell];
  unsigned boff = __float_as_uint(ce.y) + ((z >= ce.x) ? 16u : 0u);
  float4 co = *(const float4*)((const char*)tab + boff);
  float r = ((co.x * z + co.y) * z + co.z) * z + co.w;
  r = lo + r;
  return lo == kNegInf ? hi : r;
}
__device__ __forceinline__ float lseB(float sum, float x, const Tab* tab) {
  float hi = vmax(sum, x), lo = vmin(sum, x), z = hi - lo;
  bool c1 = z >= 3.3792500f;
  float tmid = c1 ? 5.789071f : 1.6320158f, tlo = c1 ? 4.426169f : 0.66153675f, thi = c1 ? 7.8162727f : 2.4912589f;
  bool c2 = z >= tmid; float t3 = c2 ? thi : tlo; bool c3 = z >= t3;
  unsigned boff = (c1 ? 64u : 0u) | (c2 ? 32u : 0u) | (c3 ? 16u : 0u);
  float4 co = *(const float4*)((const char*)tab + boff);
  float r = ((co.x * z + co.y) * z + co.z) * z + co.w;
  r = (z >= 11.862479f) ? z : r;
  r = lo + r;
  return lo == kNegInf ? hi : r;
}
// D: one LDS round trip: cell entry = {thr, coefs below, coefs above}
__device__ __forceinline__ float lseD(float sum, float x, const Tab* tab) {
  float hi = vmax(sum, x), lo = vmin(sum, x), z = hi - lo;
  int e = (int)(__float_as_uint(z) >> 20);
  int cell = min(max(e, 0x3EF), 0x418) - 0x3EF;
  const float4* f = tab->fused[cell];
  float thr = f[0].x;
  float4 a = f[1], b = f[2];
  bool up = z >= thr;
  float c0 = up ? b.x : a.x, c1 = up ? b.y : a.y, c2 = up ? b.z : a.z, c3 = up ? b.w : a.w;
  float r = ((c0 * z + c1) * z + c2) * z + c3;
  r = lo + r;
  return lo == kNegInf ? hi : r;
}
#define SEL4(c, p, q) (c ? q : p)
__device__ __forceinline__ float lseC(float sum, float x, const Tab*) {
  float hi = vmax(sum, x), lo = vmin(sum, x), z = hi - lo;
  bool c1 = z >= 3.3792500f;
  float tmid = c1 ? 5.789071f : 1.6320158f, tlo = c1 ? 4.426169f : 0.66153675f, thi = c1 ? 7.8162727f : 2.4912589f;
  float a0 = SEL4(c1, -0.0065591595f, -0.0031455354f), a1 = SEL4(c1, -0.015515756f, -0.0010110698f), a2 = SEL4(c1, -0.012890925f, -0.000196278f), a3 = SEL4(c1, -0.0072142647f, -0.0000113994f);
  float b0 = SEL4(c1, 0.12764427f, 0.046722945f), b1 = SEL4(c1, 0.14467756f, 0.018594341f), b2 = SEL4(c1, 0.13010283f, 0.0046084408f), b3 = SEL4(c1, 0.087754086f, 0.0003734731f);
  float d0 = SEL4(c1, 0.49965546f, 0.7592532f), d1 = SEL4(c1, 0.48829398f, 0.88317305f), d2 = SEL4(c1, 0.51503986f, 0.9634432f), d3 = SEL4(c1, 0.6208708f, 0.9959107f);
  float e0 = SEL4(c1, 0.6931542f, 0.43487945f), e1 = SEL4(c1, 0.6958093f, 0.25236955f), e2 = SEL4(c1, 0.6795586f, 0.09831489f), e3 = SEL4(c1, 0.5909676f, 0.0149855051f);
  bool c2 = z >= tmid; float t3 = c2 ? thi : tlo;
  float aa0 = c2 ? a2 : a0, aa1 = c2 ? a3 : a1, bb0 = c2 ? b2 : b0, bb1 = c2 ? b3 : b1, dd0 = c2 ? d2 : d0, dd1 = c2 ? d3 : d1, ee0 = c2 ? e2 : e0, ee1 = c2 ? e3 : e1;
  bool c3 = z >= t3;
  float a = c3 ? aa1 : aa0, bq = c3 ? bb1 : bb0, c = c3 ? dd1 : dd0, dq = c3 ? ee1 : ee0;
  float r = ((a * z + bq) * z + c) * z + dq;
  r = (z >= 11.862479f) ? z : r;
  r = lo + r;
  return lo == kNegInf ? hi : r;
}
// E: no LDS on the chain: the 8 lanes of a group evaluate the 8 cubic pieces of ONE chain
// speculatively (piece = lane & 7, coefficients in registers), the lane whose interval holds
// z keeps its result, an OR over the group (3 DPP steps) hands it to all 8 lanes.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_or(unsigned v) {
  return v | (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
struct PieceRegs { float c0, c1, c2, c3, tlo, thi; };
__device__ __forceinline__ PieceRegs piece_regs() {
  const int p = threadIdx.x & 7;
  PieceRegs r{kCoef[p][0], kCoef[p][1], kCoef[p][2], kCoef[p][3], p ? kBreaks[p - 1] : -1.f, kBreaks[p]};
  return r;
}
__device__ __forceinline__ float lseE(float sum, float x, const PieceRegs& P) {
  float hi = vmax(sum, x), lo = vmin(sum, x), z = hi - lo;
  float r = ((P.c0 * z + P.c1) * z + P.c2) * z + P.c3;
  r = lo + r;
  const bool sel = (z >= P.tlo) && (z < P.thi);
  unsigned v = sel ? __float_as_uint(r) : 0u;
  v = dpp_or<0xB1>(v);    // quad_perm [1,0,3,2]
  v = dpp_or<0x4E>(v);    // quad_perm [2,3,0,1]
  v = dpp_or<0x141>(v);   // row_half_mirror
  // off the chain: the identity piece (z >= 11.862479: lo + z) and the -inf operand
  const float alt = (lo == kNegInf) ? hi : lo + z;
  const bool need_alt = (lo == kNegInf) || (z >= 11.862479f);
  return need_alt ? alt : __uint_as_float(v);
}
// F: one chain per WAVE (every lane holds the same sum): the identity piece (z >= 11.862479,
// both operands finite) is a scalar branch away; otherwise E.  FL: the same with the branch
// marked likely (fast path falls through).
template <bool LIKELY>
__device__ __forceinline__ float lseF(float sum, float x, const PieceRegs& P) {
  float hi = vmax(sum, x), lo = vmin(sum, x), z = hi - lo;
  const bool far = (__float_as_uint(z) - 0x413DCCB7u) < (0x7F800000u - 0x413DCCB7u);
  const bool all_far = __ballot(far) != 0ull;
  if (LIKELY ? __builtin_expect(all_far, 1) : all_far) return lo + z;
  float r = ((P.c0 * z + P.c1) * z + P.c2) * z + P.c3;
  r = lo + r;
  const bool sel = (z >= P.tlo) && (z < P.thi);
  unsigned v = sel ? __float_as_uint(r) : 0u;
  v = dpp_or<0xB1>(v);
  v = dpp_or<0x4E>(v);
  v = dpp_or<0x141>(v);
  return (z < 11.862479f) ? __uint_as_float(v) : hi;
}
template <int V, int CH>
__global__ void __launch_bounds__(64) k(float* out, unsigned long long* cyc, int iters) {
  __shared__ Tab tab;
  load_tab(&tab);
  float s[CH];
  for (int c = 0; c < CH; c++) s[c] = threadIdx.x * 0.01f + c;
  float x = 0.3f + threadIdx.x * 0.001f;
  PieceRegs PR = piece_regs();
  if (V >= 5) {  // one chain per wave; V 5/6: every step far (x = sum - 20), V 7: every step near
    for (int c = 0; c < CH; c++) s[c] = 100.f + c;
    x = (V == 7) ? 99.f : 60.f;
  }
  if (V == 4) {  // a chain lives on 8 lanes: same operands on all of them
    for (int c = 0; c < CH; c++) s[c] = (threadIdx.x >> 3) * 0.08f + c;
    x = 0.3f + (threadIdx.x >> 3) * 0.008f;
  }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        float xx = x + (float)(u + c) * 0.37f;
        if (V == 0) s[c] = lseA(s[c], xx, &tab);
        if (V == 1) s[c] = lseB(s[c], xx, &tab);
        if (V == 2) s[c] = lseC(s[c], xx, &tab);
        if (V == 3) s[c] = lseD(s[c], xx, &tab);
        if (V == 4) s[c] = lseE(s[c], xx, PR);
        if (V == 5) s[c] = lseF<false>(s[c], xx, PR);
        if (V == 6) s[c] = lseF<true>(s[c], xx, PR);
        if (V == 7) s[c] = lseF<true>(s[c], xx, PR);
      }
    }
    x += 0.001f;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0; for (int c = 0; c < CH; c++) r += s[c];
  out[blockIdx.x * 64 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int V, int CH>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * 64 * 65536); hipMalloc(&cyc, 8 * 65536);
  const int iters = 500;
  printf("%-34s", name);
  for (int wps : {1, 2, 4, 8}) {
    int blocks = 1024 * wps;  // one wave per block -> wps waves per SIMD
    hipLaunchKernelGGL((k<V, CH>), dim3(blocks), dim3(64), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, CH>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double per = (double)h[3] / (iters * 8.0);   // cycles per step (CH folds) for one wave
    double ns_per_lse_simd = ms * 1e6 / ((double)iters * 8 * CH * wps);
    printf("  w%d: %6.0f cyc/step %5.2f ns/lse/SIMD", wps, per, ns_per_lse_simd);
  }
  printf("\n");
  hipFree(out); hipFree(cyc);
}
// does a launch that leaves most SIMDs idle run its waves at the same rate? (a lone long
// sequence keeps 200-1500 chains going on 1024 SIMDs)
template <int V>
void sparse(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * 64 * 65536); hipMalloc(&cyc, 8 * 65536);
  const int iters = 2000;
  printf("%-34s", name);
  for (int blocks : {16, 64, 192, 512, 1024}) {
    hipLaunchKernelGGL((k<V, 1>), dim3(blocks), dim3(64), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, 1>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  %4d waves: %6.2f ns/lse", blocks, ms * 1e6 / ((double)iters * 8));
  }
  printf("\n");
  hipFree(out); hipFree(cyc);
}
int main() {
  sparse<6>("FL all far, sparse launches");
  sparse<0>("A cell-LUT, sparse launches");
  run<0, 1>("A cell-LUT        1 chain");
  run<1, 1>("B ladder+LDS coef 1 chain");
  run<2, 1>("C all-VALU        1 chain");
  run<3, 1>("D fused cell entry 1 chain");
  run<3, 3>("D fused cell entry 3 chains");
  run<0, 2>("A cell-LUT        2 chains");
  run<1, 2>("B ladder+LDS coef 2 chains");
  run<2, 2>("C all-VALU        2 chains");
  run<0, 3>("A cell-LUT        3 chains");
  run<1, 3>("B ladder+LDS coef 3 chains");
  run<2, 3>("C all-VALU        3 chains");
  run<4, 1>("E 8-lane speculative 1 chain");
  run<4, 2>("E 8-lane speculative 2 chains");
  run<4, 3>("E 8-lane speculative 3 chains");
  run<5, 1>("F wave-uniform, all far      ");
  run<6, 1>("FL same, branch likely       ");
  run<7, 1>("FL wave-uniform, all near    ");
  run<6, 3>("FL all far, 3 sequential     ");
  run<0, 6>("A cell-LUT        6 chains");
  run<2, 6>("C all-VALU        6 chains");
  return 0;
}
