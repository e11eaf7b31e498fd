// Times neurst_gemm_bias_act back-to-back (device time per launch, hipEvents over many launches).
// hipcc --offload-arch=gfx950 -O2 gemm_driver.cpp -o gemm_driver -L../../neurst_amd -lneurst_hip -Wl,-rpath,'$ORIGIN/../../neurst_amd'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../include/neurst_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Shape { const char* name; int64_t M, N, K; int ta, tb, cf32, split; };
int main(int argc, char** argv) {
  std::vector<Shape> shapes = {
      {"dec q/out NT", 1280, 768, 768, 0, 1, 0, 1},   {"dec qkv NT", 1280, 2304, 768, 0, 1, 0, 1},
      {"dec ffn1 NT", 1280, 3072, 768, 0, 1, 0, 1},   {"dec ffn2 NT", 1280, 768, 3072, 0, 1, 0, 1},
      {"dec dgrad NN", 1280, 768, 768, 0, 0, 0, 1},   {"dec dgrad ffn NN", 1280, 768, 3072, 0, 0, 0, 1},
      {"dec dfc NT K=10000", 1280, 768, 10000, 0, 1, 0, 1},
      {"dec wgrad TN", 768, 768, 1280, 1, 0, 1, 1},   {"dec wgrad ffn TN", 3072, 768, 1280, 1, 0, 1, 1},
      {"enc qkv NT", 16000, 2304, 768, 0, 1, 0, 1},   {"enc ffn1 NT", 16000, 3072, 768, 0, 1, 0, 1},
      {"enc ffn2 NT", 16000, 768, 3072, 0, 1, 0, 1},  {"enc out NT", 16000, 768, 768, 0, 1, 0, 1},
      {"enc dgrad NN", 16000, 768, 2304, 0, 0, 0, 1},
      {"enc dgrad ffn2 NN", 16000, 3072, 768, 0, 0, 0, 1},    {"enc dgrad ffn1 NN", 16000, 768, 3072, 0, 0, 0, 1},
      {"enc dgrad out NN", 16000, 768, 768, 0, 0, 0, 1},      {"ctc head NN f32", 16000, 10000, 768, 0, 0, 1, 1},
      {"dec kv NT", 16000, 1536, 768, 0, 1, 0, 1},            {"dgrad ctc NT", 16000, 768, 10000, 0, 1, 0, 1},
      {"enc wgrad qkv TN s4", 2304, 768, 16000, 1, 0, 1, 4},  {"wgrad ctc TN", 768, 10000, 16000, 1, 0, 1, 1}, {"enc wgrad ffn1 TN s3", 3072, 768, 16000, 1, 0, 1, 3},
      {"enc wgrad out TN s6", 768, 768, 16000, 1, 0, 1, 6},   {"enc wgrad out TN s8", 768, 768, 16000, 1, 0, 1, 8},
      {"enc wgrad out TN s12", 768, 768, 16000, 1, 0, 1, 12},
      {"dec wgrad TN s2", 768, 768, 1280, 1, 0, 1, 2},        {"dec wgrad TN s4", 768, 768, 1280, 1, 0, 1, 4},
      {"dec wgrad qkv TN s1", 2304, 768, 1280, 1, 0, 1, 1},   {"dec wgrad qkv TN s2", 2304, 768, 1280, 1, 0, 1, 2},
      {"dec wgrad ffn TN s2", 3072, 768, 1280, 1, 0, 1, 2},   {"dec wgrad ffn TN s4", 3072, 768, 1280, 1, 0, 1, 4},
      {"dec wgrad fc TN s1", 768, 10000, 1280, 1, 0, 1, 1},   {"dec wgrad fc TN s2", 768, 10000, 1280, 1, 0, 1, 2},
  };
  size_t maxe = 16000ull * 10000;
  void *A, *B, *C; float* bias;
  CK(hipMalloc(&A, maxe * 2)); CK(hipMalloc(&B, maxe * 2)); CK(hipMalloc(&C, maxe * 4)); CK(hipMalloc(&bias, 65536));
  CK(hipMemset(A, 0x3c, maxe * 2)); CK(hipMemset(B, 0x3c, maxe * 2)); CK(hipMemset(C, 0, maxe * 4)); CK(hipMemset(bias, 0, 65536));
  // random-ish bf16 operands: fill with a pattern of varied exponents/signs
  {
    std::vector<unsigned short> h(1 << 24);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 16) & 0x807f) | 0x3f00 | ((s >> 9) & 0x0080)); }
    for (size_t off = 0; off + h.size() <= maxe; off += h.size()) {
      CK(hipMemcpy((char*)A + off * 2, h.data(), h.size() * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy((char*)B + off * 2, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& s : shapes) {
    int64_t lda = s.ta ? s.M : s.K, ldb = s.tb ? s.K : s.N;
    const int iters = 50;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) {
        int rc = neurst_gemm_bias_act(A, B, C, s.cf32 ? nullptr : bias, nullptr, s.M, s.N, s.K, lda, ldb, s.N, s.ta, s.tb, NEURST_BF16,
                                      s.cf32 ? NEURST_F32 : NEURST_BF16, 0, s.cf32, s.split, nullptr);
        if (rc) { printf("gemm error %s\n", neurst_last_error()); return 1; }
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%-24s M=%5ld N=%5ld K=%5ld  %8.1f us  %7.1f TF/s\n", s.name, (long)s.M, (long)s.N, (long)s.K, ms * 1e3 / iters,
                      2.0 * s.M * s.N * s.K / (ms * 1e-3 / iters) / 1e12);
    }
  }
  return 0;
}
