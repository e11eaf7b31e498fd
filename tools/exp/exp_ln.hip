// Experiment: where does the LN forward kernel lose its bandwidth?  hipcc --offload-arch=gfx950 -O3 exp_ln.hip -o exp_ln
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  // row_shr / row_bcast based reduction through DPP, result broadcast by readlane
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));  // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));  // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xe, true));  // row_shr:4 bank_mask e
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xc, true));  // row_shr:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true));  // row_bcast:15
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, true));  // row_bcast:31
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// MODE 0: out = x + y, yz = x + y (no stats). 1: + stats via shfl. 2: + stats via dpp.  3: like 1 but 8-byte... 
template <int MODE>
__global__ void __launch_bounds__(256) k(const bf16_t* __restrict__ x, bf16_t* __restrict__ yz, bf16_t* __restrict__ out, float* __restrict__ mean,
                                        int64_t rows, int C) {
  const int lane = threadIdx.x & 63;
  const int nchunks = C / 8;
  const float invC = 1.f / C;
  for (int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    float z[2][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int ch = lane + 64 * c;
      if (ch < nchunks) {
        int64_t off = row * C + ch * 8;
        bf16x8 a = *reinterpret_cast<const bf16x8*>(x + off), b = *reinterpret_cast<const bf16x8*>(yz + off);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] = (bf16_t)((float)a[j] + (float)b[j]); z[c][j] = (float)o[j]; s += z[c][j]; }
        *reinterpret_cast<bf16x8*>(yz + off) = o;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) z[c][j] = 0.f;
      }
    }
    float mu = 0.f, rs = 1.f;
    if (MODE == 1) {
      mu = wave_sum(s) * invC;
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c) if (lane + 64 * c < nchunks)
#pragma unroll
        for (int j = 0; j < 8; ++j) { float d = z[c][j] - mu; q += d * d; }
      rs = 1.f / sqrtf(wave_sum(q) * invC + 1e-6f);
    } else if (MODE == 2) {
      mu = wave_sum_dpp(s) * invC;
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c) if (lane + 64 * c < nchunks)
#pragma unroll
        for (int j = 0; j < 8; ++j) { float d = z[c][j] - mu; q += d * d; }
      rs = 1.f / sqrtf(wave_sum_dpp(q) * invC + 1e-6f);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int ch = lane + 64 * c;
      if (ch < nchunks) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((z[c][j] - mu) * rs);
        *reinterpret_cast<bf16x8*>(out + row * C + ch * 8) = o;
      }
    }
    if (MODE && lane == 0) mean[row] = mu;
  }
}
int main() {
  const int64_t rows = 16000; const int C = 768;
  bf16_t *x, *y, *o; float* m; char* flush;
  CK(hipMalloc(&x, rows * C * 2)); CK(hipMalloc(&y, rows * C * 2)); CK(hipMalloc(&o, rows * C * 2)); CK(hipMalloc(&m, rows * 4));
  CK(hipMalloc(&flush, 512u << 20));
  CK(hipMemset(x, 0x3c, rows * C * 2)); CK(hipMemset(y, 0x3c, rows * C * 2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode)
    for (int grid : {1024, 2048, 4000}) {
      float best = 1e9, tot = 0;
      for (int it = 0; it < 6; ++it) {
        CK(hipMemsetAsync(flush, it, 512u << 20, 0));
        CK(hipEventRecord(e0, 0));
        if (mode == 0) k<0><<<grid, 256>>>(x, y, o, m, rows, C);
        if (mode == 1) k<1><<<grid, 256>>>(x, y, o, m, rows, C);
        if (mode == 2) k<2><<<grid, 256>>>(x, y, o, m, rows, C);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it) { tot += ms; if (ms < best) best = ms; }
      }
      printf("mode %d grid %4d  min %6.1f us avg %6.1f us  %5.2f TB/s\n", mode, grid, best * 1e3, tot / 5 * 1e3, 4.0 * rows * C * 2 / (best * 1e-3) / 1e12);
    }
  return 0;
}
