// VALU issue-rate microbenchmark for gfx950: wave-instructions per cycle per SIMD
// for the op mix of the logsumexp fold, at 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void k(float* out, int iters, float seed) {
  float a[8];
  for (int x = 0; x < 8; x++) a[x] = seed + threadIdx.x * 0.001f + x;
  float c = seed * 0.5f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 8; rep++) {
#pragma unroll
      for (int x = 0; x < 8; x++) {
        if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[x]) : "v"(c));
        if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[x]) : "v"(c));
        if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[x]) : "v"(c));
        if (OP == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[x]) : "v"(c));
        if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[x]) : "v"(c));
        if (OP == 5) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(a[x]), "v"(c) : "vcc");
        if (OP == 6) asm volatile("v_cmp_ge_f32 s[10:11], %0, %1\n v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(a[x]) : "v"(c) : "s10", "s11");
        if (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[x & 6]) : "v"(*(double*)&a[(x+2)&6]));
        if (OP == 8) asm volatile("v_lshrrev_b64 %0, 2, %0" : "+v"(*(unsigned long long*)&a[x & 6]));
        if (OP == 9) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[x]) : "v"(c));
      }
    }
  }
  float s = 0;
  for (int x = 0; x < 8; x++) s += a[x];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int instr_per_iter_mult) {
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * 256 * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2000;
  printf("%-28s", name);
  for (int wps : {1, 2, 4, 8}) {
    // 256 CUs x 4 SIMDs; block = 256 threads = 4 waves -> one per SIMD; wps blocks per CU
    dim3 grid(256 * wps), block(256);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double wave_instrs_per_simd = (double)iters * 64 * instr_per_iter_mult * wps;
    double ns_per = ms * 1e6 / wave_instrs_per_simd;
    printf("  wps=%d: %.3f ns/instr/SIMD", wps, ns_per);
  }
  printf("\n");
  hipFree(out);
}

int main() {
  run<0>("v_add_f32", 1);
  run<1>("v_mul_f32", 1);
  run<2>("v_fma_f32", 1);
  run<3>("v_max_f32", 1);
  run<4>("v_cndmask_b32 vcc", 1);
  run<5>("v_cmp_ge_f32 vcc", 1);
  run<6>("v_cmp e64 + v_cndmask e64", 2);
  run<7>("v_pk_add_f32", 1);
  run<8>("v_lshrrev_b64", 1);
  run<9>("v_and_b32", 1);
  return 0;
}
