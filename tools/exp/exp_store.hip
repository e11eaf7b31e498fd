// Experiment: write bandwidth of GEMM-epilogue-like store patterns.  Output [M=16000][N] bf16; a "block" owns a 128x128 tile;
// SEG = contiguous bytes per row written by one wave-instruction (lanes*16 B / rows per instruction).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
// each block writes its 128 x 128 bf16 tile (32 KB); 4 waves
template <int SEG>   // bytes per row-segment per wave instruction: 128, 256
__global__ void __launch_bounds__(256) k(unsigned short* __restrict__ C, int64_t M, int64_t N, int tiles_n) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int64_t m0 = (int64_t)tm * 128, n0 = (int64_t)tn * 128;
  u32x4 v = {1u, 2u, 3u, (unsigned)tid};
  if constexpr (SEG == 128) {
    // current epilogue: wave (wm, wn) owns 64x64; instruction = 8 rows x 128 B
    const int wm = w >> 1, wn = w & 1;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      int64_t row = m0 + wm * 64 + p * 8 + (lane >> 3), col = n0 + wn * 64 + (lane & 7) * 8;
      if (row < M) *reinterpret_cast<u32x4*>(C + row * N + col) = v;
    }
  } else {
    // block-cooperative: instruction = 4 rows x 256 B (the full 128-column width of the tile); wave w owns rows w*32..
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      int64_t row = m0 + w * 32 + p * 4 + (lane >> 4), col = n0 + (lane & 15) * 8;
      if (row < M) *reinterpret_cast<u32x4*>(C + row * N + col) = v;
    }
  }
}
int main() {
  const int64_t M = 16000;
  unsigned short* C; CK(hipMalloc(&C, M * 3072 * 2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int64_t N : {768, 2304, 3072}) {
    int tiles_n = N / 128, tiles = 125 * tiles_n;
    for (int seg : {128, 256}) {
      float best = 1e9;
      for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 10; ++r) {
          if (seg == 128) k<128><<<tiles, 256>>>(C, M, N, tiles_n); else k<256><<<tiles, 256>>>(C, M, N, tiles_n);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        if (ms < best) best = ms;
      }
      printf("N=%5ld seg=%3d B: %7.1f us  %5.2f TB/s\n", (long)N, seg, best * 1e3, M * N * 2.0 / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
