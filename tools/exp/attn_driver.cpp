// Times the attention kernels back-to-back through the C ABI (encoder self-attention shape of the LUT step).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../include/neurst_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
extern "C" void neurst_debug_mha2_stamps(long long*);
int main(int argc, char** argv) {
  const int64_t B = 32, H = 8, T = argc > 1 ? atoi(argv[1]) : 500, dh = 96, d = H * dh;
  size_t n = B * T * 3 * d;
  unsigned short *qkv, *o, *d_o, *dqkv; float *lse, *delta; int* lens;
  CK(hipMalloc(&qkv, n * 2)); CK(hipMalloc(&dqkv, n * 2)); CK(hipMalloc(&o, n / 3 * 2)); CK(hipMalloc(&d_o, n / 3 * 2));
  CK(hipMalloc(&lse, B * H * T * 4)); CK(hipMalloc(&delta, B * H * T * 4)); CK(hipMalloc(&lens, B * 4));
  std::vector<unsigned short> h(n);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 16) & 0x807f) | 0x3f00 | ((s >> 9) & 0x0080)); }
  CK(hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_o, h.data(), n / 3 * 2, hipMemcpyHostToDevice));
  std::vector<int> hl(B, (int)T); CK(hipMemcpy(lens, hl.data(), B * 4, hipMemcpyHostToDevice));
  unsigned short* keep; float* ws;
  int64_t kbb = neurst_mha_keep_bits_bytes(B, H, T, T), wsb = neurst_mha_bwd_workspace_bytes(B, H, T, T, dh);
  CK(hipMalloc(&keep, kbb)); CK(hipMalloc(&ws, wsb ? wsb : 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (float rate : {0.0f, 0.1f}) {
    for (int which = 0; which < 5; ++which) {
      neurst_set_option("mha_fused", which >= 2 ? 2 : 0);
      neurst_set_option("mha_bwd_waves", which == 4 ? 4 : 8);
      const int iters = 20;
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) {
          int rc = (which == 0 || which == 3)
            ? neurst_mha_fwd_train(qkv, qkv + d, qkv + 2 * d, o, lse, rate > 0 ? keep : nullptr, 3 * d, 3 * d, 3 * d, d, B, H, T, T, dh, NEURST_MASK_KEYPAD, lens, rate, 123u, NEURST_BF16, nullptr)
            : neurst_mha_bwd_train_ws(qkv, qkv + d, qkv + 2 * d, o, d_o, lse, rate > 0 ? keep : nullptr, delta, dqkv, dqkv + d, dqkv + 2 * d, 3 * d, 3 * d, 3 * d, d, 3 * d, 3 * d, 3 * d,
                             B, H, T, T, dh, NEURST_MASK_KEYPAD, lens, rate, 123u, NEURST_BF16, ws, wsb, nullptr);
          if (rc) { printf("error %s\n", neurst_last_error()); return 1; }
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      double fl = 4.0 * B * H * T * T * dh * ((which == 1 || which == 2 || which == 4) ? 2.5 : 1.0);
      printf("T=%ld rate=%.1f %s %8.1f us  %6.1f TF/s\n", (long)T, rate, which == 4 ? "bwd(fused,4 waves)" : which == 3 ? "fwd(wide)  " : which == 2 ? "bwd(fused,8 waves)" : which ? "bwd(dq+dkv)" : "fwd        ", ms * 1e3 / iters, fl / (ms * 1e-3 / iters) / 1e12);
    }
  }
  for (int abl : {0, 1, 2, 4, 8, 15}) {   // timing-only ablations of the wide forward
    neurst_set_option("mha_fused", 2);
    neurst_set_option("mha_fwd_ablate", abl);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 20; ++i)
        neurst_mha_fwd_train(qkv, qkv + d, qkv + 2 * d, o, lse, nullptr, 3 * d, 3 * d, 3 * d, d, B, H, T, T, dh, NEURST_MASK_KEYPAD, lens, 0.f, 123u, NEURST_BF16, nullptr);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("fwd(wide) ablate %2d: %8.1f us\n", abl, ms * 1e3 / 20);
  }
  neurst_set_option("mha_fwd_ablate", 0);
  for (int waves : {4, 8}) {   // phase stamps of workgroup 0 of the fused backward (cycles)
    neurst_set_option("mha_bwd_waves", waves);
    long long* st; CK(hipMalloc(&st, 64 * 8)); CK(hipMemset(st, 0, 64 * 8));
    neurst_debug_mha2_stamps(st);
    neurst_set_option("mha_fused", 2);
    int rc = neurst_mha_bwd_train_ws(qkv, qkv + d, qkv + 2 * d, o, d_o, lse, keep, delta, dqkv, dqkv + d, dqkv + 2 * d, 3 * d, 3 * d, 3 * d, d, 3 * d, 3 * d, 3 * d,
                             B, H, T, T, dh, NEURST_MASK_KEYPAD, lens, 0.1f, 123u, NEURST_BF16, ws, wsb, nullptr);
    if (rc) { printf("error %s\n", neurst_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    long long hs[64]; CK(hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost));
    neurst_debug_mha2_stamps(nullptr);
    printf("stamps, %d waves (delta cycles):", waves);
    for (int i = 1; i < 64 && hs[i]; ++i) printf(" %lld", hs[i] - hs[i - 1]);
    printf("\n");
  }
  return 0;
}
