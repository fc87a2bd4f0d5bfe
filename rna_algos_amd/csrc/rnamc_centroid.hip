// rnamc_centroid.hip — the Theta(n^3) fill of the gamma-centroid fold on the GPU, for several
// thresholds at once (reference: src/centroid_fold.rs:35-63, run for 18 gammas by
// src/bin/centroid_fold.rs:147-161).
//
//   M[i][j] = max( M[i+1][j], M[i][j-1], M[i+1][j-1] + gamma * p(i,j) - 1   (if (i,j) has a bpp),
//                  max_{i<k<j} M[i][k] + M[k+1][j] ),     M = 0 below and on the main diagonal.
//
// (max,+) is order-free in f32: every candidate is ONE rounded addition (the pair term one
// multiply, one add, one subtract, never fused: this file is compiled with -ffp-contract=off)
// and the maximum of a set does not depend on the order it is taken in.  So any tiling of the
// k-reduction gives the reference's bits, and the traceback — which compares floats for exact
// equality (src/centroid_fold.rs:66-102) — runs on the host off the matrices this kernel leaves.
//
// One launch per anti-diagonal; blockIdx.x = cell group, blockIdx.y = threshold.  M is kept
// row-major AND column-major so that both factors of the bifurcation term are contiguous in k.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "rnamc_device.h"

namespace rnamc {

namespace {

template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dppf(float old, float x) {
  return __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_update_dpp(
      static_cast<int>(__float_as_uint(old)), static_cast<int>(__float_as_uint(x)), CTRL, ROWMASK, 0xF,
      false)));
}
__device__ __forceinline__ float wave_fmax(float v) {
  const float lo = -__builtin_inff();
  v = fmaxf(v, dppf<0x111, 0xF>(lo, v));
  v = fmaxf(v, dppf<0x112, 0xF>(lo, v));
  v = fmaxf(v, dppf<0x114, 0xF>(lo, v));
  v = fmaxf(v, dppf<0x118, 0xF>(lo, v));
  v = fmaxf(v, dppf<0x142, 0xA>(lo, v));
  v = fmaxf(v, dppf<0x143, 0xC>(lo, v));
  return __uint_as_float(static_cast<uint32_t>(
      __builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(v)), 63)));
}

template <int TPC>
__global__ void __launch_bounds__(256) k_centroid(CentroidBatch a, uint32_t d) {
  __shared__ float red[4];
  const uint32_t n = a.n, ld = a.ld;
  const uint32_t i = blockIdx.x * (256 / TPC) + threadIdx.x / TPC;
  if (i + d >= n) return;
  const uint32_t j = i + d, t = threadIdx.x % TPC;
  float* __restrict__ mr = a.m + static_cast<size_t>(blockIdx.y) * 2u * a.msz;  // [r * ld + c]
  float* __restrict__ mc = mr + a.msz;                                          // [c * ld + r]
  const size_t row_i = static_cast<size_t>(i) * ld, col_j = static_cast<size_t>(j) * ld;
  // bifurcations: k = i+1 .. j-1, M[i][k] + M[k+1][j]
  float best = 0.f;  // (every M is >= 0: the empty structure)
  const float* __restrict__ A = mr + row_i + i + 1;
  const float* __restrict__ B = mc + col_j + i + 2;
  for (uint32_t k = t; k + 1 < d; k += 4u * TPC) {
    const uint32_t k1 = k + TPC, k2 = k + 2u * TPC, k3 = k + 3u * TPC;
    const float c0 = A[k] + B[k];
    const float c1 = k1 + 1 < d ? A[k1] + B[k1] : 0.f;
    const float c2 = k2 + 1 < d ? A[k2] + B[k2] : 0.f;
    const float c3 = k3 + 1 < d ? A[k3] + B[k3] : 0.f;
    best = fmaxf(fmaxf(best, fmaxf(c0, c1)), fmaxf(c2, c3));
  }
  best = wave_fmax(best);
  if (TPC > 64) {
    if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = best;
    __syncthreads();
    best = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
  if (t != 0u) return;
  best = fmaxf(best, mr[row_i + ld + j]);  // M[i+1][j]
  best = fmaxf(best, mr[row_i + j - 1]);   // M[i][j-1]
  const float pr = a.bpp[static_cast<size_t>(d) * n - (static_cast<size_t>(d) * (d - 1u)) / 2u + i];
  if (pr >= -0.5f) {  // present in the SparseProbMat
    const float g = a.gammas[blockIdx.y];
    const float cand = mr[row_i + ld + j - 1] + g * pr - 1.f;  // ((M + g*p) - 1), as the reference parses it
    best = fmaxf(best, cand);
  }
  mr[row_i + j] = best;
  mc[col_j + i] = best;
}

}  // namespace

void launch_centroid(const CentroidBatch& a, uint32_t d, uint32_t n_gammas, hipStream_t st) {
  const uint32_t cells = a.n - d;
  // few cells with long sums: a whole workgroup per cell
  if (static_cast<uint64_t>(cells) * n_gammas >= 2048u || d < 512u)
    hipLaunchKernelGGL(k_centroid<64>, dim3((cells + 3) / 4, n_gammas, 1), dim3(256), 0, st, a, d);
  else
    hipLaunchKernelGGL(k_centroid<256>, dim3(cells, n_gammas, 1), dim3(256), 0, st, a, d);
}

}  // namespace rnamc
