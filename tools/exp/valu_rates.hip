// Issue cost of a few vector instructions on one wave per SIMD (cycles per instruction, independent streams of 8 registers).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
template <int OP> __global__ void k(unsigned* out, long long* cyc, unsigned seed) {
  unsigned x[8];
  for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 8 + i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(0x85EBCA6Bu));
        if (OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(0x85EBCAu));
        if (OP == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(0x85EBCA6Bu));
        if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(0x85EBCAu));
        if (OP == 5) asm volatile("v_lshrrev_b32 %0, 13, %0" : "+v"(x[i]));
        if (OP == 6) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(0x85EBCA6Bu));
        if (OP == 7) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
        if (OP == 8) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(x[i]));
        if (OP == 9) asm volatile("v_bfe_i32 %0, %0, %1, 1" : "+v"(x[i]) : "v"(threadIdx.x & 31));
        if (OP == 10) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(x[i]) : "v"(seed));
        if (OP == 11) asm volatile("v_cndmask_b32_e64 %0, %1, %0, s[10:11]" : "+v"(x[i]) : "v"(seed) : "s10", "s11");
        if (OP == 12) asm volatile("v_cmp_ge_u32 vcc, %0, %1" : : "v"(x[i]), "v"(seed) : "vcc");
        if (OP == 13) asm volatile("v_cmp_ge_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %1, %0, vcc" : "+v"(x[i]) : "v"(seed) : "vcc");
        if (OP == 14) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(x[i]) : "v"(seed) : "vcc");
        if (OP == 15) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  unsigned s = 0;
  for (int i = 0; i < 8; ++i) s ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[OP] = t1 - t0;
}
int main() {
  unsigned* out; long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 32 * 8); hipMemset(cyc, 0, 256);
  const char* names[] = {"v_mul_lo_u32", "v_mul_u32_u24", "v_xor_b32", "v_exp_f32", "v_mad_u32_u24", "v_lshrrev_b32", "v_mul_hi_u32", "v_fma_f32", "v_cndmask_b32 0", "v_bfe_i32", "v_cndmask vcc", "v_cndmask sgpr", "v_cmp_ge_u32", "v_cmp+v_cndmask", "v_sub_co_u32", "v_max_f32"};
  for (int waves = 1; waves <= 2; ++waves) {
    // 256 * waves threads per block = `waves` waves per SIMD on the one CU that runs block 0
    k<0><<<256, 256 * waves>>>(out, cyc, 1); k<1><<<256, 256 * waves>>>(out, cyc, 1); k<2><<<256, 256 * waves>>>(out, cyc, 1); k<3><<<256, 256 * waves>>>(out, cyc, 1);
    k<4><<<256, 256 * waves>>>(out, cyc, 1); k<5><<<256, 256 * waves>>>(out, cyc, 1); k<6><<<256, 256 * waves>>>(out, cyc, 1); k<7><<<256, 256 * waves>>>(out, cyc, 1);
    k<8><<<256, 256 * waves>>>(out, cyc, 1); k<9><<<256, 256 * waves>>>(out, cyc, 1);
    k<10><<<256, 256 * waves>>>(out, cyc, 1); k<11><<<256, 256 * waves>>>(out, cyc, 1); k<12><<<256, 256 * waves>>>(out, cyc, 1);
    k<13><<<256, 256 * waves>>>(out, cyc, 1); k<14><<<256, 256 * waves>>>(out, cyc, 1); k<15><<<256, 256 * waves>>>(out, cyc, 1);
    hipDeviceSynchronize();
    long long h[32]; hipMemcpy(h, cyc, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("%d wave(s)/SIMD  %-16s %.2f cycles per wave-instruction\n", waves, names[i], (double)h[i] / (16.0 * REP * 8));
  }
  return 0;
}
