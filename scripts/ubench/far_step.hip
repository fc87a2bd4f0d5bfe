// Cost of one sure-far fold step  sum = x + (sum - x)  of a wave-uniform chain on gfx950, by the
// way the per-step operand x reaches the chain (one wave, nothing else on the chip):
//  V0: v_readlane inside the loop (what the compiler makes of the plain loop)
//  V1: the same, the read of step l+1 issued before the two operations of step l
//  V2: 8 lane reads issued first, then the 16 dependent operations
//  V3: no operand traffic at all (x in an SGPR): the floor of the dependent sub+add pair
//  V4: chain held in lane 0 only, operands rotated towards it (v_mov_dpp wave_ror:1)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o far_step far_step.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float lane_val(float v, unsigned lane) {
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), (int)lane));
}

template <int V>
__global__ void __launch_bounds__(64) k(const float* in, float* out, unsigned long long* ticks, int iters, float xs) {
  const unsigned lane = threadIdx.x;
  const float t = in[lane];
  float sum = in[64];
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
    if (V == 0) {
      for (unsigned l = 0; l < 64; l++) {
        const float x = lane_val(t, l);
        const float z = sum - x;
        sum = x + z;
      }
    } else if (V == 1) {
      float x = lane_val(t, 0);
#pragma unroll 1
      for (unsigned l = 0; l < 64; l++) {
        const float xn = lane_val(t, (l + 1) & 63u);
        const float z = sum - x;
        sum = x + z;
        x = xn;
      }
    } else if (V == 2) {
#pragma unroll 1
      for (unsigned l = 0; l < 64; l += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = lane_val(t, l + u);
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const float z = sum - x[u];
          sum = x[u] + z;
        }
      }
    } else if (V == 3) {
#pragma unroll 8
      for (unsigned l = 0; l < 64; l++) {
        const float z = sum - xs;
        sum = xs + z;
        asm volatile("" : "+v"(sum));
      }
    } else if (V == 4) {
      float r = t;  // lane 0 holds the operand of the current step
#pragma unroll 8
      for (unsigned l = 0; l < 64; l++) {
        const float z = sum - r;
        sum = r + z;
        // wave_ror:1 moves lane l+1's value to lane l ... (0x13C = wave rotate right by 1:
        // lane i reads lane i-1); rotate the other way by reading from the next lane: wave_rol:1 = 0x134
        r = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(r), 0x134, 0xF, 0xF, false));
      }
    }
  }
  const unsigned long long t1 = wall_clock64();
  out[blockIdx.x * 64 + lane] = sum;
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name, const float* d_in, float* d_out, unsigned long long* d_t) {
  const int iters = 2000;
  hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, d_in, d_out, d_t, 10, -3.f);
  hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, d_in, d_out, d_t, iters, -3.f);
  hipDeviceSynchronize();
  unsigned long long t = 0;
  float o[64];
  hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost);
  hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
  printf("%-44s %7.2f ns/step   (sum lane0 %.6f lane63 %.6f)\n", name, t * 10.0 / (iters * 64.0), o[0], o[63]);
}

int main() {
  std::vector<float> h(65);
  for (int i = 0; i < 64; i++) h[i] = -40.f - 0.37f * i;
  h[64] = 5.25f;
  float *d_in, *d_out;
  unsigned long long* d_t;
  hipMalloc(&d_in, 65 * 4);
  hipMalloc(&d_out, 64 * 4);
  hipMalloc(&d_t, 8);
  hipMemcpy(d_in, h.data(), 65 * 4, hipMemcpyHostToDevice);
  run<0>("V0 readlane in the loop", d_in, d_out, d_t);
  run<1>("V1 readlane one step ahead", d_in, d_out, d_t);
  run<2>("V2 8 readlanes, then 16 dependent ops", d_in, d_out, d_t);
  run<3>("V3 operand in an SGPR (floor)", d_in, d_out, d_t);
  run<4>("V4 chain in lane 0, operands rotated (dpp)", d_in, d_out, d_t);
  return 0;
}
