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
        if (OP == 10) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[x]) : "v"(c));
        if (OP == 11) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[x]));
        if (OP == 12) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(a[x]) : "v"(c));
        if (OP == 13) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[x]) : "v"(c));
        if (OP == 14) asm volatile("v_mul_f32_e64 %0, |%0|, %1" : "+v"(a[x]) : "v"(c));
        if (OP == 15) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[x]) : "v"(c));
        if (OP == 16) asm volatile("v_med3_u32 %0, %0, %1, %1" : "+v"(a[x]) : "v"(c));
        if (OP == 17) asm volatile("v_bfe_u32 %0, %0, 3, 11" : "+v"(a[x]));
        if (OP == 18) asm volatile("v_cmp_ge_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[x]) : "v"(c) : "vcc");
        if (OP == 19) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(*(unsigned long long*)&a[x & 6]) : "v"(*(unsigned long long*)&a[(x + 2) & 6]));
        if (OP == 20) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[x]) : "v"(c));
        if (OP == 21) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[x]) : "v"(c));
        if (OP == 22) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[x]));
        if (OP == 23) asm volatile("v_sub_f32_e64 %0, %1, |%0|" : "+v"(a[x]) : "v"(c));
        if (OP == 24) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[x]) : "v"(c));
        if (OP == 25) asm volatile("v_cmp_ge_f32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[x]) : "v"(c) : "vcc");
        if (OP == 26) asm volatile("v_add_f32 %0, %0, %1\n v_cmp_ge_f32 vcc, %0, %1\n v_add_f32 %0, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[x]) : "v"(c) : "vcc");
        if (OP == 27) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[x]) : "v"(c));
        if (OP == 28) asm volatile("v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(a[x]) : "v"(c));
        if (OP == 29) asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_max_f32 %0, %0, %1" : "+v"(a[x]) : "v"(c));
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
  run<10>("v_sub_f32", 1);
  run<11>("v_lshrrev_b32", 1);
  run<12>("v_lshl_add_u32", 1);
  run<13>("v_and_or_b32", 1);
  run<14>("v_mul_f32_e64 |abs|", 1);
  run<15>("v_min_f32", 1);
  run<16>("v_med3_u32", 1);
  run<17>("v_bfe_u32", 1);
  run<18>("v_cmp vcc + v_cndmask vcc", 2);
  run<19>("v_lshl_add_u64", 1);
  run<20>("v_add_u32", 1);
  run<21>("v_bfi_b32", 1);
  run<22>("v_ashrrev_i32", 1);
  run<23>("v_sub_f32_e64 |abs|", 1);
  run<24>("v_max_u32", 1);
  run<25>("v_cmp vcc + v_addc_co vcc", 2);
  run<26>("add, cmp vcc, add, cndmask vcc", 4);
  run<27>("v_xor_b32", 1);
  run<28>("v_cndmask_b32_sdwa", 1);
  run<29>("3 x v_add_f32 + v_max_f32", 4);
  return 0;
}
