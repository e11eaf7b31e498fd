// Experiment: Adam arena update bandwidth vs unroll / grid.  120.3 M parameters.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int U>
__global__ void __launch_bounds__(256) adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                            __bf16* __restrict__ shadow, int64_t n4, float lr_t, float b1, float b2, float eps, float gs) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i0 < n4; i0 += stride * U) {
    f32x4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n4) { pp[u] = ((f32x4*)p)[i]; gg[u] = ((const f32x4*)g)[i]; mm[u] = ((f32x4*)m)[i]; vv[u] = ((f32x4*)v)[i]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i >= n4) break;
      bf16x4 sh;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gj = gg[u][j] * gs;
        mm[u][j] = b1 * mm[u][j] + (1.f - b1) * gj;
        vv[u][j] = b2 * vv[u][j] + (1.f - b2) * gj * gj;
        pp[u][j] = pp[u][j] - lr_t * mm[u][j] / (sqrtf(vv[u][j]) + eps);
        sh[j] = (__bf16)pp[u][j];
      }
      ((f32x4*)p)[i] = pp[u]; ((f32x4*)m)[i] = mm[u]; ((f32x4*)v)[i] = vv[u]; ((bf16x4*)shadow)[i] = sh;
    }
  }
}
int main() {
  const int64_t n = 120283136, n4 = n / 4;
  float *p, *g, *m, *v; __bf16* s;
  CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&s, n * 2));
  CK(hipMemset(p, 0, n * 4)); CK(hipMemset(g, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int U : {1, 2, 4})
    for (int grid : {2048, 4096, 8192, 16384, 65536}) {
      float best = 1e9;
      for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0, 0));
        if (U == 1) adam<1><<<grid, 256>>>(p, g, m, v, s, n4, 1e-3f, .9f, .98f, 1e-9f, 1.f);
        if (U == 2) adam<2><<<grid, 256>>>(p, g, m, v, s, n4, 1e-3f, .9f, .98f, 1e-9f, 1.f);
        if (U == 4) adam<4><<<grid, 256>>>(p, g, m, v, s, n4, 1e-3f, .9f, .98f, 1e-9f, 1.f);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("U=%d grid=%6d  %7.1f us  %5.2f TB/s\n", U, grid, best * 1e3, 30.0 * n / (best * 1e-3) / 1e12);
    }
  return 0;
}
