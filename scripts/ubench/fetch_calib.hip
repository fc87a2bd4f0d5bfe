// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the two access patterns of the sweep
// kernels, with known byte counts (each byte is read exactly once, footprint >> caches):
//   pattern 0: one dword per lane, 256 B contiguous per wave instruction, rows 8 KB apart
//   pattern 1: every lane streams its own 64-byte pieces (4 x dwordx4 per 16 steps)
//   pattern 2: float4 per lane streaming copy-like read (the guide's reference pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ void __launch_bounds__(256) k_dword_rows(const float* __restrict__ a, float* out,
                                                    uint32_t stride, uint32_t steps) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
  const uint32_t per_row = stride / 64;
  const size_t base = (size_t)(wave / per_row) * steps * stride + (wave % per_row) * 64 + lane;
  float acc = 0;
  for (uint32_t t = 0; t < steps; t += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = a[base + (size_t)(t + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; u++) acc += v[u];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) k_lane_pieces(const float4* __restrict__ a, float* out,
                                                     uint32_t col_len4, uint32_t steps16) {
  // lane owns one column of col_len4 float4; walks it 64 bytes at a time
  const size_t lane_col = (size_t)(blockIdx.x * blockDim.x + threadIdx.x);
  const float4* col = a + lane_col * col_len4;
  float acc = 0;
  for (uint32_t s = 0; s < steps16; s++) {
    float4 v0 = col[4 * s], v1 = col[4 * s + 1], v2 = col[4 * s + 2], v3 = col[4 * s + 3];
    acc += v0.x + v0.y + v0.z + v0.w + v1.x + v1.y + v1.z + v1.w + v2.x + v2.y + v2.z + v2.w +
           v3.x + v3.y + v3.z + v3.w;
  }
  out[lane_col] = acc;
}

__global__ void __launch_bounds__(256) k_float4_stream(const float4* __restrict__ a, float* out,
                                                       size_t n4) {
  float acc = 0;
  for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < n4;
       x += (size_t)gridDim.x * blockDim.x) {
    float4 v = a[x];
    acc += v.x + v.y + v.z + v.w;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  const int pattern = argc > 1 ? atoi(argv[1]) : 0;
  const size_t bytes = 16ull << 30;  // 16 GiB read once
  float *a, *out;
  hipMalloc(&a, bytes + 4096);
  const size_t out_elems = 16ull << 20;  // >= threads of the largest launch (8.4 M)
  hipMalloc(&out, out_elems * 4);
  hipMemset(a, 0, bytes);
  hipDeviceSynchronize();
  if (pattern == 0) {
    const uint32_t stride = 2048, steps = 512;  // a wave reads 512 rows x 256 B
    const size_t waves = bytes / (256ull * steps);
    if (waves * 64 > out_elems || (waves / 32) * steps * (size_t)stride * 4 > bytes) return 3;
    hipLaunchKernelGGL(k_dword_rows, dim3(waves / 4), dim3(256), 0, 0, a, out, stride, steps);
    printf("pattern 0 dword rows: %.3f GB named\n", waves * 256.0 * steps / 1e9);
  } else if (pattern == 1) {
    const uint32_t col_len4 = 512;  // 8 KB columns
    const size_t lanes = bytes / (col_len4 * 16ull);
    if (lanes > out_elems || lanes * col_len4 * 16ull > bytes) return 3;
    hipLaunchKernelGGL(k_lane_pieces, dim3(lanes / 256), dim3(256), 0, 0, (const float4*)a, out,
                       col_len4, col_len4 / 4);
    printf("pattern 1 lane pieces: %.3f GB named\n", lanes * col_len4 * 16.0 / 1e9);
  } else {
    hipLaunchKernelGGL(k_float4_stream, dim3(4096), dim3(256), 0, 0, (const float4*)a, out, bytes / 16);
    printf("pattern 2 float4 stream: %.3f GB named\n", bytes / 1e9);
  }
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  return e == hipSuccess ? 0 : 4;
}
