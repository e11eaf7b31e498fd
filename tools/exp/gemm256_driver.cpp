// Checks the 256 x 256-tile GEMM kernel against the 128 x 128 kernels (bit-exact on small-integer operands: every partial
// sum is an exact fp32 integer, so the summation order cannot matter) and times both back-to-back on random operands.
// hipcc --offload-arch=gfx950 -O2 gemm256_driver.cpp -o gemm256_driver -L../../neurst_amd -lneurst_hip -Wl,-rpath,'$ORIGIN/../../neurst_amd'
//   gemm256_driver check | time | all
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/neurst_hip.h"
extern "C" void neurst_debug_gemm256_stamps(unsigned long long* buf);
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
struct Shape { const char* name; int64_t M, N, K; int ta, tb, cf32, bias, relu, gate, acc; };

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "all";
  const size_t maxe = 16000ull * 10000;
  void *A, *B, *C0, *C1, *G; float* bias;
  CK(hipMalloc(&A, maxe * 2)); CK(hipMalloc(&B, maxe * 2)); CK(hipMalloc(&C0, maxe * 4)); CK(hipMalloc(&C1, maxe * 4)); CK(hipMalloc(&G, maxe * 2));
  CK(hipMalloc(&bias, 65536 * 4));
  int bad = 0;
  if (!strcmp(mode, "check") || !strcmp(mode, "all")) {
    // integers in [-3, 3]: |sum| <= 9 K < 2^24 for K <= 16000 x ...; bias integers; gate signs
    std::vector<unsigned short> h(1 << 22);
    unsigned s = 777;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = f2bf((float)((int)((s >> 20) % 7) - 3)); }
    for (size_t off = 0; off < maxe; off += h.size()) {
      size_t n = maxe - off < h.size() ? maxe - off : h.size();
      CK(hipMemcpy((char*)A + off * 2, h.data(), n * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy((char*)G + off * 2, h.data() + 1, (n - 1) * 2, hipMemcpyHostToDevice));
    }
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = f2bf((float)((int)((s >> 20) % 5) - 2)); }
    for (size_t off = 0; off < maxe; off += h.size()) {
      size_t n = maxe - off < h.size() ? maxe - off : h.size();
      CK(hipMemcpy((char*)B + off * 2, h.data(), n * 2, hipMemcpyHostToDevice));
    }
    std::vector<float> hb(65536);
    for (auto& v : hb) { s = s * 1664525u + 1013904223u; v = (float)((int)((s >> 20) % 9) - 4); }
    CK(hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    std::vector<Shape> shapes = {
        {"NT 512x512x128", 512, 512, 128, 0, 1, 0, 0, 0, 0, 0},
        {"NT 256x256x64", 256, 256, 64, 0, 1, 1, 0, 0, 0, 0},
        {"NT ragged 300x520x200 bias relu", 300, 520, 200, 0, 1, 0, 1, 1, 0, 0},
        {"NT ragged f32 acc 777x264x72", 777, 264, 72, 0, 1, 1, 1, 0, 0, 1},
        {"NT K odd tiles 1000x768x448", 1000, 768, 448, 0, 1, 0, 1, 0, 0, 0},
        {"NT K tail 1030x776x1000", 1030, 776, 1000, 0, 1, 0, 0, 0, 0, 0},
        {"NN 1000x768x448 gate", 1000, 768, 448, 0, 0, 0, 0, 0, 1, 0},
        {"NN K tail 520x264x1000", 520, 264, 1000, 0, 0, 1, 0, 0, 0, 0},
        {"TN 768x520x1000 f32 acc", 768, 520, 1000, 1, 0, 1, 0, 0, 0, 1},
        {"TN 264x776x520", 264, 776, 520, 1, 0, 0, 1, 0, 0, 0},
        {"TT 520x264x456", 520, 264, 456, 1, 1, 0, 0, 0, 0, 0},
        {"NT enc qkv", 16000, 2304, 768, 0, 1, 0, 1, 0, 0, 0},
        {"NT enc ffn1 relu", 16000, 3072, 768, 0, 1, 0, 1, 1, 0, 0},
        {"NT enc ffn2", 16000, 768, 3072, 0, 1, 0, 1, 0, 0, 0},
        {"NN enc dgrad ffn2 gate", 16000, 3072, 768, 0, 0, 0, 0, 0, 1, 0},
        {"NN enc dgrad qkv", 16000, 768, 2304, 0, 0, 0, 0, 0, 0, 0},
        {"NN ctc head f32", 16000, 10000, 768, 0, 0, 1, 0, 0, 0, 0},
        {"NT dgrad ctc K=10000", 16000, 768, 10000, 0, 1, 0, 0, 0, 0, 0},
        {"TN wgrad-like 3072x768x16000 f32 acc", 3072, 768, 16000, 1, 0, 1, 0, 0, 0, 1},
    };
    neurst_set_option("gemm256_min_tiles", 1);
    for (int bn : {256, 192})
    for (auto& sh : shapes) {
      neurst_set_option("gemm256_bn", bn);
      const int64_t lda = sh.ta ? sh.M : sh.K, ldb = sh.tb ? sh.K : sh.N, ldc = sh.N;
      const size_t cbytes = (size_t)sh.M * sh.N * (sh.cf32 ? 4 : 2);
      void* Cs[2] = {C0, C1};
      for (int which = 0; which < 2; ++which) {
        neurst_set_option("gemm256", which);
        CK(hipMemset(Cs[which], sh.acc ? 0x00 : 0xff, cbytes));      // accumulate: start from zeros; else poison
        if (sh.acc) { float one = 0; (void)one; }
        for (int rep = 0; rep < (which ? 3 : 1); ++rep) {            // the new kernel three times (races show as differences)
          if (rep) CK(hipMemset(Cs[which], sh.acc ? 0x00 : 0xff, cbytes));
          int rc = neurst_gemm_bias_act(A, B, Cs[which], sh.bias ? bias : nullptr, sh.gate ? G : nullptr, sh.M, sh.N, sh.K, lda, ldb, ldc, sh.ta,
                                        sh.tb, NEURST_BF16, sh.cf32 ? NEURST_F32 : NEURST_BF16, sh.relu, sh.acc, 1, nullptr);
          if (rc) { printf("gemm error %s\n", neurst_last_error()); return 1; }
          CK(hipDeviceSynchronize());
          if (which) {
            std::vector<unsigned char> h0(cbytes), h1(cbytes);
            CK(hipMemcpy(h0.data(), C0, cbytes, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h1.data(), C1, cbytes, hipMemcpyDeviceToHost));
            size_t ndiff = 0, first = 0;
            const size_t es = sh.cf32 ? 4 : 2;
            for (size_t i = 0; i < cbytes; i += es)
              if (memcmp(&h0[i], &h1[i], es)) { if (!ndiff) first = i / es; ++ndiff; }
            printf("%-40s bn %d rep %d: %s (%zu of %zu differ%s)\n", sh.name, bn, rep, ndiff ? "MISMATCH" : "ok", ndiff, cbytes / es, "");
            if (ndiff) {
              ++bad;
              printf("   first at row %zu col %zu\n", first / sh.N, first % sh.N);
              size_t rows_bad[8] = {0}; (void)rows_bad;
            }
          }
        }
      }
    }
    printf(bad ? "CHECK FAILED (%d)\n" : "CHECK OK\n", bad);
  }
  if (!strcmp(mode, "time") || !strcmp(mode, "all")) {
    std::vector<unsigned short> h(1 << 24);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 16) & 0x807f) | 0x3f00 | ((s >> 9) & 0x0080)); }
    for (size_t off = 0; off + h.size() <= maxe; off += h.size()) {
      CK(hipMemcpy((char*)A + off * 2, h.data(), h.size() * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy((char*)B + off * 2, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    CK(hipMemset(bias, 0, 65536 * 4));
    std::vector<Shape> shapes = {
        {"enc qkv NT", 16000, 2304, 768, 0, 1, 0, 1, 0, 0, 0},        {"enc ffn1 NT relu", 16000, 3072, 768, 0, 1, 0, 1, 1, 0, 0},
        {"enc ffn2 NT", 16000, 768, 3072, 0, 1, 0, 1, 0, 0, 0},       {"enc out NT", 16000, 768, 768, 0, 1, 0, 1, 0, 0, 0},
        {"dec kv NT", 16000, 1536, 768, 0, 1, 0, 1, 0, 0, 0},         {"enc fc NT K=480", 16000, 768, 480, 0, 1, 0, 0, 0, 0, 0},
        {"enc dgrad qkv NN", 16000, 768, 2304, 0, 0, 0, 0, 0, 0, 0},  {"enc dgrad ffn2 NN gate", 16000, 3072, 768, 0, 0, 0, 0, 0, 1, 0},
        {"enc dgrad ffn1 NN", 16000, 768, 3072, 0, 0, 0, 0, 0, 0, 0}, {"enc dgrad out NN", 16000, 768, 768, 0, 0, 0, 0, 0, 0, 0},
        {"ctc head NN f32", 16000, 10000, 768, 0, 0, 1, 0, 0, 0, 0},  {"dgrad ctc NT K=10000", 16000, 768, 10000, 0, 1, 0, 0, 0, 0, 0},
        {"costt ffn1 NT", 16000, 2048, 512, 0, 1, 0, 1, 1, 0, 0},     {"costt qkv NT", 16000, 1536, 512, 0, 1, 0, 1, 0, 0, 0},
        {"square 8192 NT", 8192, 8192, 8192, 0, 1, 0, 0, 0, 0, 0},    {"square 4096 NT", 4096, 4096, 4096, 0, 1, 0, 0, 0, 0, 0},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    neurst_set_option("gemm256_min_tiles", 1);
    for (auto& sh : shapes) {
      const int64_t lda = sh.ta ? sh.M : sh.K, ldb = sh.tb ? sh.K : sh.N, ldc = sh.N;
      float us[3] = {0, 0, 0};
      for (int round = 0; round < 3; ++round)
        for (int which = 0; which < 3; ++which) {
          neurst_set_option("gemm256", which ? 1 : 0);
          neurst_set_option("gemm256_bn", which == 2 ? 192 : 256);
          const int iters = 20;
          CK(hipEventRecord(e0, 0));
          for (int i = 0; i < iters; ++i) {
            int rc = neurst_gemm_bias_act(A, B, C0, sh.bias ? bias : nullptr, sh.gate ? G : nullptr, sh.M, sh.N, sh.K, lda, ldb, ldc, sh.ta, sh.tb,
                                          NEURST_BF16, sh.cf32 ? NEURST_F32 : NEURST_BF16, sh.relu, sh.acc, 1, nullptr);
            if (rc) { printf("gemm error %s\n", neurst_last_error()); return 1; }
          }
          CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (round) us[which] = (round == 1 ? ms * 1e3f / iters : (us[which] < ms * 1e3f / iters ? us[which] : ms * 1e3f / iters));
        }
      const double fl = 2.0 * sh.M * sh.N * sh.K;
      printf("%-24s %5ldx%5ldx%5ld  128-tile %7.1f us %6.1f TF | 256x256 %7.1f us %6.1f TF (x%.2f) | 256x192 %7.1f us %6.1f TF (x%.2f)\n", sh.name,
             (long)sh.M, (long)sh.N, (long)sh.K, us[0], fl / us[0] / 1e6, us[1], fl / us[1] / 1e6, us[0] / us[1], us[2], fl / us[2] / 1e6, us[0] / us[2]);
    }
    neurst_set_option("gemm256_bn", 0);
  }
  if (!strcmp(mode, "wgrad") || !strcmp(mode, "all")) {
    // weight gradient + bias gradient: whole-K 256 x 192 tiles (few workgroups, no split-K) against the 128-tile split-K kernel
    struct W { const char* name; int64_t M, N, K; int sk; };
    std::vector<W> ws = {{"wgrad qkv", 2304, 768, 16000, 4}, {"wgrad out", 768, 768, 16000, 8}, {"wgrad ffn1", 3072, 768, 16000, 3},
                         {"wgrad ffn2", 768, 3072, 16000, 3}, {"wgrad ragged", 776, 520, 1000, 2}};
    float* opsum0; float* opsum1; CK(hipMalloc(&opsum0, 65536 * 4)); CK(hipMalloc(&opsum1, 65536 * 4));
    {
      std::vector<unsigned short> h(1 << 22);
      unsigned s = 4242;
      for (auto& v : h) { s = s * 1664525u + 1013904223u; v = f2bf((float)((int)((s >> 20) % 5) - 2)); }
      for (size_t off = 0; off < maxe; off += h.size()) {
        size_t n = maxe - off < h.size() ? maxe - off : h.size();
        CK(hipMemcpy((char*)A + off * 2, h.data(), n * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy((char*)B + off * 2 , h.data() + 3, (n - 3) * 2, hipMemcpyHostToDevice));
      }
    }
    neurst_set_option("gemm256_min_tiles", 1);
    for (auto& w : ws) {
      void* Cs[2] = {C0, C1}; float* os[2] = {opsum0, opsum1};
      for (int which = 0; which < 2; ++which) {
        neurst_set_option("gemm256", which);
        for (int rep = 0; rep < (which ? 2 : 1); ++rep) {
          CK(hipMemset(Cs[which], 0, w.M * w.N * 4)); CK(hipMemset(os[which], 0, w.M * 4));
          int rc = neurst_gemm_wgrad_bias(A, B, (float*)Cs[which], os[which], 1, w.M, w.N, w.K, w.M, w.N, w.N, NEURST_BF16, w.sk, nullptr);
          if (rc) { printf("wgrad error %s\n", neurst_last_error()); return 1; }
          CK(hipDeviceSynchronize());
          if (which) {
            std::vector<float> h0(w.M * w.N), h1(w.M * w.N), o0(w.M), o1(w.M);
            CK(hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(o0.data(), opsum0, w.M * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), opsum1, w.M * 4, hipMemcpyDeviceToHost));
            size_t nd = 0, ns = 0;
            for (size_t i = 0; i < h0.size(); ++i) nd += h0[i] != h1[i];
            for (size_t i = 0; i < o0.size(); ++i) ns += o0[i] != o1[i];
            printf("%-14s rep %d: C %s (%zu differ), bias sums %s (%zu differ; e.g. %g vs %g)\n", w.name, rep, nd ? "MISMATCH" : "ok", nd, ns ? "MISMATCH" : "ok", ns,
                   o0[5], o1[5]);
            if (nd || ns) ++bad;
          }
        }
      }
    }
    // timing on random operands: one launch alone, and the four gradients of an encoder block on four streams at once
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
    hipStream_t st[4]; for (auto& x : st) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (int which = 0; which < 2; ++which) {
      neurst_set_option("gemm256", which);
      for (int wi = 0; wi < 4; ++wi) {
        auto& w = ws[wi];
        float best = 1e9;
        for (int round = 0; round < 3; ++round) {
          CK(hipEventRecord(e0, st[0]));
          for (int i = 0; i < 5; ++i) neurst_gemm_wgrad_bias(A, B, (float*)C0, opsum0, 1, w.M, w.N, w.K, w.M, w.N, w.N, NEURST_BF16, w.sk, st[0]);
          CK(hipEventRecord(e1, st[0])); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms / 5 < best) best = ms / 5;
        }
        printf("%s %-12s alone: %8.1f us  %7.1f TF/s\n", which ? "256x192 whole-K" : "128-tile split-K", w.name, best * 1e3, 2.0 * w.M * w.N * w.K / best / 1e9);
      }
      // a block's four gradients, 4 rounds: serial on one stream vs one stream per gradient
      for (int conc = 0; conc < 2; ++conc) {
        float best = 1e9;
        for (int round = 0; round < 3; ++round) {
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0, st[0]));
          for (int x = 1; x < 4; ++x) { hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, st[0])); CK(hipStreamWaitEvent(st[x], ev, 0)); CK(hipEventDestroy(ev)); }
          for (int i = 0; i < 4; ++i)
            for (int wi = 0; wi < 4; ++wi) {
              auto& w = ws[wi];
              neurst_gemm_wgrad_bias(A, B, (float*)C0 + (size_t)wi * 3072 * 3072, opsum0 + wi * 4096, 1, w.M, w.N, w.K, w.M, w.N, w.N, NEURST_BF16, w.sk, st[conc ? wi : 0]);
            }
          for (int x = 1; x < 4; ++x) { hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, st[x])); CK(hipStreamWaitEvent(st[0], ev, 0)); CK(hipEventDestroy(ev)); }
          CK(hipEventRecord(e1, st[0])); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double fl = 4 * 2.0 * 16000 * (2304.0 * 768 + 768.0 * 768 + 2 * 3072.0 * 768);
        printf("%s, 4 blocks x 4 gradients, %s: %8.1f us  %7.1f TF/s\n", which ? "256x192 whole-K" : "128-tile split-K", conc ? "four streams" : "one stream", best * 1e3,
               fl / best / 1e9);
      }
    }
  }
  if (!strcmp(mode, "ablate")) {
    // timing-only builds of the K loop (results are wrong): which part of a phase sets its length?
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {
      std::vector<unsigned short> h(1 << 24);
      unsigned s = 12345;
      for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 16) & 0x807f) | 0x3f00 | ((s >> 9) & 0x0080)); }
      for (size_t off = 0; off + h.size() <= maxe; off += h.size()) {
        CK(hipMemcpy((char*)A + off * 2, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy((char*)B + off * 2, h.data(), h.size() * 2, hipMemcpyHostToDevice));
      }
    }
    neurst_set_option("gemm256_min_tiles", 1); neurst_set_option("gemm256", 1); neurst_set_option("gemm256_bn", 256);
    const int masks[] = {0, 64, 1, 4, 8, 16};
    for (int m : masks) {
      neurst_set_option("gemm256_ablate", m);
      float best = 1e9;
      for (int round = 0; round < 3; ++round) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) neurst_gemm_bias_act(A, B, C0, nullptr, nullptr, 8192, 8192, 8192, 8192, 8192, 8192, 0, 1, NEURST_BF16, NEURST_BF16, 0, 0, 1, nullptr);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 10 < best) best = ms / 10;
      }
      printf("ablate %2d (1 no DMA, 4 no MFMA, 8 no 2nd barrier, 16 no setprio, 64 DMA between the MFMAs): 8192^3 %8.1f us = %6.0f cycles/K-tile at 2.0 GHz\n", m, best * 1e3,
             best * 1e-3 / (4 * 128) * 2.0e9);
    }
    neurst_set_option("gemm256_ablate", 0);
  }
  if (!strcmp(mode, "stamps") || !strcmp(mode, "all")) {
    // where one workgroup's time goes (diagnostic instantiation): cycles between the stamps, averaged over workgroups
    unsigned long long* st; CK(hipMalloc(&st, 256 * 24 * 8));
    std::vector<Shape> shapes = {{"enc qkv NT", 16000, 2304, 768, 0, 1, 0, 1, 0, 0, 0}, {"enc ffn2 NT", 16000, 768, 3072, 0, 1, 0, 1, 0, 0, 0},
                                 {"enc ffn1 NT", 16000, 3072, 768, 0, 1, 0, 1, 1, 0, 0}};
    neurst_set_option("gemm256_min_tiles", 1); neurst_set_option("gemm256", 1);
    for (int bn : {256, 192})
    for (auto& sh : shapes) {
      neurst_set_option("gemm256_bn", bn);
      printf("bn %d ", bn);
      std::vector<unsigned long long> h(256 * 24);
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(st, 0, 256 * 24 * 8));
        neurst_debug_gemm256_stamps(rep == 2 ? st : nullptr);
        for (int i = 0; i < (rep == 2 ? 1 : 5); ++i)
          neurst_gemm_bias_act(A, B, C0, bias, nullptr, sh.M, sh.N, sh.K, sh.K, sh.K, sh.N, 0, 1, NEURST_BF16, NEURST_BF16, sh.relu, 0, 1, nullptr);
        CK(hipDeviceSynchronize());
      }
      neurst_debug_gemm256_stamps(nullptr);
      CK(hipMemcpy(h.data(), st, 256 * 24 * 8, hipMemcpyDeviceToHost));
      printf("%s: per workgroup (cycles; mean over workgroups with that many tiles)\n", sh.name);
      unsigned long long tmin = ~0ull, tmax = 0;
      for (int w = 0; w < 256; ++w) if (h[w * 24]) { if (h[w * 24] < tmin) tmin = h[w * 24]; if (h[w * 24 + 21] > tmax) tmax = h[w * 24 + 21]; }
      double clk = 0; int nclk = 0;
      for (int w = 0; w < 256; ++w) if (h[w * 24 + 22] > h[w * 24 + 23]) { clk += (double)(h[w * 24 + 21] - h[w * 24]) / (double)(h[w * 24 + 22] - h[w * 24 + 23]) * 100.0; ++nclk; }
      printf("  kernel span %llu cycles, clock ~%.0f MHz\n", tmax - tmin, nclk ? clk / nclk : 0.0);
      for (int ntile = 1; ntile <= 4; ++ntile) {
        double acc[24] = {0}; int cnt = 0;
        for (int w = 0; w < 256; ++w) {
          const unsigned long long* r = &h[w * 24];
          if (!r[0]) continue;
          int nt = 0; while (nt < 4 && r[2 + nt * 4]) ++nt;
          if (nt != ntile) continue;
          ++cnt;
          acc[0] += (double)(r[1] - r[0]);
          for (int t = 0; t < nt; ++t) {
            acc[1 + t * 4] += (double)(r[3 + t * 4] - r[2 + t * 4]);   // K loop
            acc[2 + t * 4] += (double)(r[4 + t * 4] - r[3 + t * 4]);   // epilogue issue
            acc[3 + t * 4] += (double)(r[5 + t * 4] - r[4 + t * 4]);   // drain
            acc[4 + t * 4] += (double)((t + 1 < nt ? r[2 + (t + 1) * 4] : r[21]) - r[5 + t * 4]);   // to next loop / exit
          }
          acc[20] += (double)(r[0] - tmin);
          acc[21] += (double)(r[21] - r[0]);
        }
        if (!cnt) continue;
        printf("  %d workgroups with %d tiles: start offset %.0f, prologue %.0f, total %.0f\n", cnt, ntile, acc[20] / cnt, acc[0] / cnt, acc[21] / cnt);
        for (int t = 0; t < ntile; ++t)
          printf("    tile %d: K loop %.0f  epilogue %.0f  drain %.0f  gap %.0f\n", t, acc[1 + t * 4] / cnt, acc[2 + t * 4] / cnt, acc[3 + t * 4] / cnt, acc[4 + t * 4] / cnt);
      }
    }
  }
  return bad ? 2 : 0;
}
