// Bandwidth-bound elementwise / gather kernels of the hot path (HBM roofline).
// a1 splice+downsample, a3 specaugment, a5 timing+dropout, a6 row mask, a15 decoder
// inputs, a16 embedding, a20 Adam, dtype casts, a21 greedy-step bookkeeping.
#include <stdarg.h>

#include "common.h"

// ---------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";
void neurst_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* neurst_last_error(void) { return g_err; }
extern "C" int neurst_abi_version(void) { return NEURST_ABI_VERSION; }

static inline int grid_for(int64_t n, int block, int per_thread = 1) {
  int64_t g = cdiv64(n, (int64_t)block * per_thread);
  if (g > 8192) g = 8192;  // grid-stride beyond 32 blocks per CU
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------- a1
// out[b,t',c*F+f] = x[b, rate*t'+c, f] if rate*t'+c < len_raw[b] else 0
template <typename TO>
__global__ void splice_kernel(const float* __restrict__ x, const int32_t* __restrict__ len_raw, TO* __restrict__ out,
                              int32_t* __restrict__ len_out, int64_t B, int64_t Traw, int64_t Tout, int64_t F, int nctx,
                              int rate) {
  const int64_t W = F * (nctx + 1);
  const int64_t total = B * Tout * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t w = i % W, r = i / W;
    int64_t t = r % Tout, b = r / Tout;
    int64_t c = w / F, f = w % F;
    int64_t src = t * rate + c;
    int len = len_raw[b];
    float v = (src < len && src < Traw) ? x[(b * Traw + src) * F + f] : 0.f;
    out[i] = from_f32<TO>(v);
  }
  if (blockIdx.x == 0)
    for (int64_t b = threadIdx.x; b < B; b += blockDim.x) len_out[b] = (len_raw[b] + rate - 1) / rate;
}

extern "C" int neurst_splice_downsample(const float* x, const int32_t* len_raw, void* out, int32_t* len_out, int64_t B,
                                        int64_t Traw, int64_t F, int num_context, int rate, int out_dtype, void* stream) {
  NEURST_CHECK_ARG(B > 0 && Traw > 0 && F > 0 && num_context >= 0 && rate >= 1, "splice_downsample: bad shape");
  int64_t Tout = cdiv64(Traw, rate);
  int64_t total = B * Tout * F * (num_context + 1);
  int grid = grid_for(total, 256);
  if (out_dtype == NEURST_BF16)
    splice_kernel<bf16_t><<<grid, 256, 0, as_stream(stream)>>>(x, len_raw, (bf16_t*)out, len_out, B, Traw, Tout, F,
                                                               num_context, rate);
  else
    splice_kernel<float><<<grid, 256, 0, as_stream(stream)>>>(x, len_raw, (float*)out, len_out, B, Traw, Tout, F,
                                                              num_context, rate);
  NEURST_CHECK_LAUNCH("splice_downsample");
  return NEURST_OK;
}

// ---------------------------------------------------------------- a3
// one block per utterance: count non-zero frames, then draw the 8 parameters
__global__ void __launch_bounds__(1024) specaug_params_kernel(const float* __restrict__ x, int32_t* __restrict__ params, int64_t T, int64_t v,
                                      int F, int Tmax, float p, uint32_t key) {
  __shared__ float red[16];
  const int64_t b = blockIdx.x;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  float cnt = 0.f;
  // a frame counts when sum |x| != 0, i.e. when any element is non-zero (NaN included): the wave stops at the first 64
  // features that hold one, so real speech costs 256 B per frame instead of the whole row (87 -> ~15 us at 32 x 500 x 480)
  for (int64_t t = w; t < T; t += nw) {
    const float* row = x + (b * T + t) * v;
    bool any = false;
    for (int64_t f0 = 0; f0 < v && !any; f0 += 64) {
      const int64_t f = f0 + lane;
      any = __any(f < v && !(row[f] == 0.f));
    }
    if (lane == 0 && any) cnt += 1.f;
  }
  float tot = block_sum(cnt, red);
  if (threadIdx.x == 0) {
    int tau = (int)tot;
    uint32_t r[8];
    uint32_t k2 = fmix32(key);
    for (int j = 0; j < 8; ++j) r[j] = fmix32((uint32_t)(8 * b + j) ^ k2);
    auto randint = [](uint32_t rr, int maxval) { return (int)(rr % (uint32_t)(maxval < 1 ? 1 : maxval)); };
    int f = randint(r[0], F), f0 = randint(r[1], (int)v - f);
    int f2 = randint(r[2], F), f02 = randint(r[3], (int)v - f2);
    int t = randint(r[4], Tmax);
    int cap = (int)(p * (float)tau);
    t = min(t, cap);
    int t0 = randint(r[5], tau - t);
    int t2 = randint(r[6], Tmax);
    t2 = max(min(t2 + t, cap) - t, 0);
    int t02 = randint(r[7], tau - t2);
    int32_t* o = params + b * 8;
    o[0] = f; o[1] = f0; o[2] = f2; o[3] = f02; o[4] = t; o[5] = t0; o[6] = t2; o[7] = t02;
  }
}

__global__ void specaug_apply_kernel(float* __restrict__ x, const int32_t* __restrict__ params, int64_t B, int64_t T,
                                     int64_t v) {
  const int64_t total = B * T * v;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t f = i % v, r = i / v;
    int64_t t = r % T, b = r / T;
    const int32_t* p = params + b * 8;
    bool z = (f >= p[1] && f < p[1] + p[0]) || (f >= p[3] && f < p[3] + p[2]) || (t >= p[5] && t < p[5] + p[4]) ||
             (t >= p[7] && t < p[7] + p[6]);
    if (z) x[i] = 0.f;
  }
}

extern "C" int neurst_specaug_params(const float* x, int32_t* params, int64_t B, int64_t T, int64_t v, int F, int Tmax,
                                     float p, uint32_t key, void* stream) {
  NEURST_CHECK_ARG(B > 0 && T > 0 && v > 0 && F >= 1 && Tmax >= 1, "specaug_params: bad shape");
  specaug_params_kernel<<<(int)B, 1024, 0, as_stream(stream)>>>(x, params, T, v, F, Tmax, p, key);   // 16 waves share one utterance's frames
  NEURST_CHECK_LAUNCH("specaug_params");
  return NEURST_OK;
}
extern "C" int neurst_specaug_apply(float* x, const int32_t* params, int64_t B, int64_t T, int64_t v, void* stream) {
  NEURST_CHECK_ARG(B > 0 && T > 0 && v > 0, "specaug_apply: bad shape");
  specaug_apply_kernel<<<grid_for(B * T * v, 256), 256, 0, as_stream(stream)>>>(x, params, B, T, v);
  NEURST_CHECK_LAUNCH("specaug_apply");
  return NEURST_OK;
}

// ---------------------------------------------------------------- a5 + dropout
template <typename TI, typename TO>
__global__ void timing_dropout_kernel(const TI* __restrict__ x, const float* __restrict__ sig, TO* __restrict__ y,
                                      int64_t total, int64_t TC, uint32_t thr, float scale, uint32_t key) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float v = to_f32<TI>(x[i]) + sig[i % TC];
    if (thr) v = dropout_keep(key, (uint64_t)i, thr) ? v * scale : 0.f;
    y[i] = from_f32<TO>(v);
  }
}

extern "C" int neurst_add_timing_dropout_fwd(const void* x, int x_dtype, const float* signal, void* y, int y_dtype,
                                             int64_t B, int64_t T, int64_t C, float rate, uint32_t key, void* stream) {
  NEURST_CHECK_ARG(B > 0 && T > 0 && C > 0 && rate >= 0.f && rate < 1.f, "add_timing_dropout: bad args");
  int64_t total = B * T * C;
  uint32_t thr = rate > 0.f ? rate_threshold(rate) : 0u;
  float sc = keep_scale(rate);
  int grid = grid_for(total, 256, 4);
  hipStream_t s = as_stream(stream);
#define LAUNCH(TI, TO) \
  timing_dropout_kernel<TI, TO><<<grid, 256, 0, s>>>((const TI*)x, signal, (TO*)y, total, T * C, thr, sc, key)
  if (x_dtype == NEURST_F32 && y_dtype == NEURST_F32) LAUNCH(float, float);
  else if (x_dtype == NEURST_F32 && y_dtype == NEURST_BF16) LAUNCH(float, bf16_t);
  else if (x_dtype == NEURST_BF16 && y_dtype == NEURST_BF16) LAUNCH(bf16_t, bf16_t);
  else LAUNCH(bf16_t, float);
#undef LAUNCH
  NEURST_CHECK_LAUNCH("add_timing_dropout");
  return NEURST_OK;
}

template <typename T>
__global__ void dropout_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int64_t n, uint32_t thr, float scale,
                                   uint32_t key) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = to_f32<T>(dy[i]);
    if (thr) v = dropout_keep(key, (uint64_t)i, thr) ? v * scale : 0.f;
    dx[i] = from_f32<T>(v);
  }
}
extern "C" int neurst_dropout_bwd(const void* dy, void* dx, int64_t n, int dtype, float rate, uint32_t key, void* stream) {
  NEURST_CHECK_ARG(n > 0 && rate >= 0.f && rate < 1.f, "dropout_bwd: bad args");
  uint32_t thr = rate > 0.f ? rate_threshold(rate) : 0u;
  int grid = grid_for(n, 256, 4);
  if (dtype == NEURST_BF16)
    dropout_bwd_kernel<bf16_t><<<grid, 256, 0, as_stream(stream)>>>((const bf16_t*)dy, (bf16_t*)dx, n, thr, keep_scale(rate), key);
  else
    dropout_bwd_kernel<float><<<grid, 256, 0, as_stream(stream)>>>((const float*)dy, (float*)dx, n, thr, keep_scale(rate), key);
  NEURST_CHECK_LAUNCH("dropout_bwd");
  return NEURST_OK;
}

// ---------------------------------------------------------------- dst += src
template <typename T>
__global__ void add_inplace_kernel(T* __restrict__ dst, const T* __restrict__ src, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = from_f32<T>(to_f32<T>(dst[i]) + to_f32<T>(src[i]));
}
extern "C" int neurst_add_inplace(void* dst, const void* src, int64_t n, int dtype, void* stream) {
  NEURST_CHECK_ARG(dst && src && n > 0, "add_inplace: bad args");
  int grid = grid_for(n, 256, 4);
  if (dtype == NEURST_BF16) add_inplace_kernel<bf16_t><<<grid, 256, 0, as_stream(stream)>>>((bf16_t*)dst, (const bf16_t*)src, n);
  else add_inplace_kernel<float><<<grid, 256, 0, as_stream(stream)>>>((float*)dst, (const float*)src, n);
  NEURST_CHECK_LAUNCH("add_inplace");
  return NEURST_OK;
}

// ---------------------------------------------------------------- a4 lengths
__global__ void lens_ceil_div_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, int64_t B, int div) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < B) out[i] = (in[i] + div - 1) / div;
}
extern "C" int neurst_lens_ceil_div(const int32_t* in, int32_t* out, int64_t B, int div, void* stream) {
  NEURST_CHECK_ARG(in && out && B > 0 && div > 0, "lens_ceil_div: bad args");
  lens_ceil_div_kernel<<<(int)cdiv64(B, 256), 256, 0, as_stream(stream)>>>(in, out, B, div);
  NEURST_CHECK_LAUNCH("lens_ceil_div");
  return NEURST_OK;
}

// ---------------------------------------------------------------- a6
template <typename T>
__global__ void mask_rows_kernel(T* __restrict__ x, const int32_t* __restrict__ lens, int64_t B, int64_t Tn, int64_t C) {
  const int64_t total = B * Tn * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / C;
    int64_t t = r % Tn, b = r / Tn;
    if (t >= lens[b]) x[i] = from_f32<T>(0.f);
  }
}
extern "C" int neurst_mask_rows(void* x, const int32_t* lens, int64_t B, int64_t T, int64_t C, int dtype, void* stream) {
  NEURST_CHECK_ARG(B > 0 && T > 0 && C > 0, "mask_rows: bad shape");
  int grid = grid_for(B * T * C, 256, 4);
  if (dtype == NEURST_BF16) mask_rows_kernel<bf16_t><<<grid, 256, 0, as_stream(stream)>>>((bf16_t*)x, lens, B, T, C);
  else mask_rows_kernel<float><<<grid, 256, 0, as_stream(stream)>>>((float*)x, lens, B, T, C);
  NEURST_CHECK_LAUNCH("mask_rows");
  return NEURST_OK;
}

// ---------------------------------------------------------------- a15
__global__ void build_dec_input_kernel(const int32_t* __restrict__ labels, const int32_t* __restrict__ lens,
                                       int32_t* __restrict__ inp, int64_t B, int64_t L, int32_t sos) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B * L) return;
  int64_t l = i % L, b = i / L;
  int32_t v = (l == 0) ? sos : labels[i - 1];
  inp[i] = (l < lens[b]) ? v : 0;
}
extern "C" int neurst_build_decoder_input(const int32_t* labels, const int32_t* len_labels, int32_t* inp, int64_t B,
                                          int64_t L, int32_t sos, void* stream) {
  NEURST_CHECK_ARG(B > 0 && L > 0, "build_decoder_input: bad shape");
  build_dec_input_kernel<<<(int)cdiv64(B * L, 256), 256, 0, as_stream(stream)>>>(labels, len_labels, inp, B, L, sos);
  NEURST_CHECK_LAUNCH("build_decoder_input");
  return NEURST_OK;
}

// ---------------------------------------------------------------- a16 embedding
template <typename TO>
__global__ void embedding_fwd_kernel(const int32_t* __restrict__ ids, const float* __restrict__ table,
                                     const float* __restrict__ sig, TO* __restrict__ out, int64_t BL, int64_t L,
                                     int64_t D, int64_t V, uint32_t thr, float scale, uint32_t key) {
  const int64_t total = BL * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t d = i % D, r = i / D;
    int64_t l = r % L;
    int32_t id = ids[r];
    id = id < 0 ? 0 : (id >= V ? (int32_t)V - 1 : id);
    float e = table[(int64_t)id * D + d];
    float v = e + (e + sig[l * D + d]);   // transformer_decoder.py:195  x += x + signal
    if (thr) v = dropout_keep(key, (uint64_t)i, thr) ? v * scale : 0.f;
    out[i] = from_f32<TO>(v);
  }
}
extern "C" int neurst_embedding_fwd(const int32_t* ids, const float* table, const float* signal, void* out, int64_t B,
                                    int64_t L, int64_t D, int64_t V, float rate, uint32_t key, int out_dtype, void* stream) {
  NEURST_CHECK_ARG(B > 0 && L > 0 && D > 0 && V > 0 && rate >= 0.f && rate < 1.f, "embedding_fwd: bad args");
  uint32_t thr = rate > 0.f ? rate_threshold(rate) : 0u;
  int grid = grid_for(B * L * D, 256);
  if (out_dtype == NEURST_BF16)
    embedding_fwd_kernel<bf16_t><<<grid, 256, 0, as_stream(stream)>>>(ids, table, signal, (bf16_t*)out, B * L, L, D, V, thr, keep_scale(rate), key);
  else
    embedding_fwd_kernel<float><<<grid, 256, 0, as_stream(stream)>>>(ids, table, signal, (float*)out, B * L, L, D, V, thr, keep_scale(rate), key);
  NEURST_CHECK_LAUNCH("embedding_fwd");
  return NEURST_OK;
}

template <typename T>
__global__ void embedding_bwd_kernel(const int32_t* __restrict__ ids, const T* __restrict__ dout, float* __restrict__ dtable,
                                     int64_t BL, int64_t D, int64_t V, uint32_t thr, float scale, uint32_t key) {
  const int64_t total = BL * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t d = i % D, r = i / D;
    int32_t id = ids[r];
    id = id < 0 ? 0 : (id >= V ? (int32_t)V - 1 : id);
    float g = to_f32<T>(dout[i]);
    if (thr) g = dropout_keep(key, (uint64_t)i, thr) ? g * scale : 0.f;
    atomicAdd(dtable + (int64_t)id * D + d, 2.f * g);
  }
}
extern "C" int neurst_embedding_bwd(const int32_t* ids, const void* dout, float* dtable, int64_t B, int64_t L, int64_t D,
                                    int64_t V, float rate, uint32_t key, int dtype, void* stream) {
  NEURST_CHECK_ARG(B > 0 && L > 0 && D > 0 && V > 0 && rate >= 0.f && rate < 1.f, "embedding_bwd: bad args");
  uint32_t thr = rate > 0.f ? rate_threshold(rate) : 0u;
  int grid = grid_for(B * L * D, 256);
  if (dtype == NEURST_BF16)
    embedding_bwd_kernel<bf16_t><<<grid, 256, 0, as_stream(stream)>>>(ids, (const bf16_t*)dout, dtable, B * L, D, V, thr, keep_scale(rate), key);
  else
    embedding_bwd_kernel<float><<<grid, 256, 0, as_stream(stream)>>>(ids, (const float*)dout, dtable, B * L, D, V, thr, keep_scale(rate), key);
  NEURST_CHECK_LAUNCH("embedding_bwd");
  return NEURST_OK;
}

// ---------------------------------------------------------------- a20 Adam (flat arena, 16 B per lane)
// EPT 16-byte elements per thread (all loads issued before the first use); NT: streaming hints - every arena is touched once
// per step and is larger than the Infinity Cache, so keeping its lines helps nobody
template <int EPT, bool NTEMP>
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            bf16_t* __restrict__ shadow, int64_t n, float lr_t, float b1, float b2, float eps, float gs,
                            const int32_t* __restrict__ skip) {
  if (skip && *skip) return;     // guarded update (neurst_adam_step_guarded): a device-side veto, the same for every workgroup
  const int64_t n4 = n >> 2;
  auto ld = [](const float* q, int64_t i) __attribute__((always_inline)) {
    const f32x4* a = reinterpret_cast<const f32x4*>(q) + i;
    if constexpr (NTEMP) return __builtin_nontemporal_load(a);
    else return *a;
  };
  auto st = [](float* q, int64_t i, f32x4 x) __attribute__((always_inline)) {
    f32x4* a = reinterpret_cast<f32x4*>(q) + i;
    if constexpr (NTEMP) __builtin_nontemporal_store(x, a);
    else *a = x;
  };
  const int64_t i0 = (blockIdx.x * (int64_t)blockDim.x) * EPT + threadIdx.x;
  f32x4 pp[EPT], gg[EPT], mm[EPT], vv[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = i0 + (int64_t)e * blockDim.x;
    if (i < n4) { pp[e] = ld(p, i); gg[e] = ld(g, i); mm[e] = ld(m, i); vv[e] = ld(v, i); }
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int64_t i = i0 + (int64_t)e * blockDim.x;
    if (i >= n4) break;
    bf16x4 sh;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gj = gg[e][j] * gs;
      mm[e][j] = b1 * mm[e][j] + (1.f - b1) * gj;
      vv[e][j] = b2 * vv[e][j] + (1.f - b2) * gj * gj;
      pp[e][j] = pp[e][j] - lr_t * mm[e][j] / (sqrtf(vv[e][j]) + eps);
      sh[j] = (bf16_t)pp[e][j];
    }
    st(p, i, pp[e]);
    st(m, i, mm[e]);
    st(v, i, vv[e]);
    if (shadow) reinterpret_cast<bf16x4*>(shadow)[i] = sh;      // read again by the next step's first kernels: no hint
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    int64_t i = (n4 << 2) + threadIdx.x;
    float gj = g[i] * gs;
    float mj = b1 * m[i] + (1.f - b1) * gj;
    float vj = b2 * v[i] + (1.f - b2) * gj * gj;
    float pj = p[i] - lr_t * mj / (sqrtf(vj) + eps);
    p[i] = pj; m[i] = mj; v[i] = vj;
    if (shadow) shadow[i] = (bf16_t)pj;
  }
}
extern "C" int neurst_adam_step_guarded(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr_t,
                                        float beta1, float beta2, float eps, float grad_scale, const int32_t* skip_flag, void* stream);
extern "C" int neurst_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr_t,
                                float beta1, float beta2, float eps, float grad_scale, void* stream) {
  return neurst_adam_step_guarded(p, g, m, v, shadow_bf16, n, lr_t, beta1, beta2, eps, grad_scale, nullptr, stream);
}
extern "C" int neurst_adam_step_guarded(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr_t,
                                        float beta1, float beta2, float eps, float grad_scale, const int32_t* skip_flag, void* stream) {
  NEURST_CHECK_ARG(n > 0, "adam_step: n <= 0");
  NEURST_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: arenas must be 16-byte aligned");
  NEURST_CHECK_ARG(((uintptr_t)shadow_bf16 & 7) == 0, "adam_step: shadow must be 8-byte aligned");
  // Two 16-byte elements per thread with streaming hints, no grid-stride loop.  Measured on the LUT arenas (120.3 M
  // parameters): 578 us = 6.2 TB/s; one element per thread 681 us with or without the hints, two without them 681, three
  // or four with them 584; an 8192-block persistent grid 4.5 TB/s.  NEURST_ADAM_VARIANT=0: one element, no hints (A/B timing).
  static const int variant = [] { const char* e = getenv("NEURST_ADAM_VARIANT"); return e ? atoi(e) : 2; }();
  const int ept = variant == 0 ? 1 : 2;
  int64_t ablocks = cdiv64(n >> 2, 256 * ept);
  NEURST_CHECK_ARG(ablocks < (int64_t)1 << 31, "adam_step: n too large for one launch");
  if (ablocks < 1) ablocks = 1;
#define ADAM(EPT, NTEMP) adam_kernel<EPT, NTEMP><<<(unsigned)ablocks, 256, 0, as_stream(stream)>>>(p, g, m, v, (bf16_t*)shadow_bf16, n, lr_t, beta1, beta2, eps, grad_scale, skip_flag)
  if (variant == 0) ADAM(1, false);
  else ADAM(2, true);
#undef ADAM
  NEURST_CHECK_LAUNCH("adam_step");
  return NEURST_OK;
}

__global__ void cast_f2b_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (bf16_t)s[i];
}
__global__ void cast_b2f_kernel(const bf16_t* __restrict__ s, float* __restrict__ d, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (float)s[i];
}
// handle_gradients (st/tools/tftools/gradientTools.py:40-67): the L2 penalty is y += alpha x on a variable's slice of the
// gradient arena, the global norm a sum of squares over the whole arena (block partials, summed by the caller in fp64)
__global__ void axpy_f32_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = fmaf(alpha, x[i], y[i]);
}
extern "C" int neurst_axpy_f32(float* y, const float* x, float alpha, int64_t n, void* stream) {
  NEURST_CHECK_ARG(y && x && n > 0, "axpy: bad args");
  axpy_f32_kernel<<<grid_for(n, 256, 4), 256, 0, as_stream(stream)>>>(y, x, alpha, n);
  NEURST_CHECK_LAUNCH("axpy_f32");
  return NEURST_OK;
}
extern "C" int64_t neurst_sumsq_blocks(int64_t n) { const int64_t g = cdiv64(n, 4096); return g > 1024 ? 1024 : (g < 1 ? 1 : g); }
__global__ void __launch_bounds__(256) sumsq_f32_kernel(const float* __restrict__ x, double* __restrict__ partial, int64_t n) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += (double)x[i] * (double)x[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
extern "C" int neurst_sumsq_f32(const float* x, int64_t n, void* partial_f64, void* stream) {
  NEURST_CHECK_ARG(x && partial_f64 && n > 0, "sumsq: bad args");
  sumsq_f32_kernel<<<(int)neurst_sumsq_blocks(n), 256, 0, as_stream(stream)>>>(x, (double*)partial_f64, n);
  NEURST_CHECK_LAUNCH("sumsq_f32");
  return NEURST_OK;
}

extern "C" int neurst_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  NEURST_CHECK_ARG(n > 0, "cast: n <= 0");
  cast_f2b_kernel<<<grid_for(n, 256, 4), 256, 0, as_stream(stream)>>>(src, (bf16_t*)dst, n);
  NEURST_CHECK_LAUNCH("cast_f32_to_bf16");
  return NEURST_OK;
}
extern "C" int neurst_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
  NEURST_CHECK_ARG(n > 0, "cast: n <= 0");
  cast_b2f_kernel<<<grid_for(n, 256, 4), 256, 0, as_stream(stream)>>>((const bf16_t*)src, dst, n);
  NEURST_CHECK_LAUNCH("cast_bf16_to_f32");
  return NEURST_OK;
}

// ---------------------------------------------------------------- a21 greedy step
// one block per utterance: first-maximum argmax over V, then the bookkeeping of
// transformer_decoder.py:83-90
__global__ void greedy_step_kernel(const float* __restrict__ logits, int32_t* __restrict__ preds, int32_t* __restrict__ finished,
                                   int32_t* __restrict__ len_decoded, int32_t* __restrict__ n_unfinished, int64_t V,
                                   int64_t step, int64_t max_steps, int32_t eos) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int64_t b = blockIdx.x;
  const float* row = logits + b * V;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int64_t i = threadIdx.x; i < V; i += blockDim.x) {
    float x = row[i];
    if (x > best || (x == best && (int)i < bi)) { best = x; bi = (int)i; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    float ob = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sv[w] = best; si[w] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i)
      if (sv[i] > best || (sv[i] == best && si[i] < bi)) { best = sv[i]; bi = si[i]; }
    preds[b * max_steps + step] = bi;
    int f = finished[b] | (bi == eos);
    finished[b] = f;
    len_decoded[b] += 1 - f;
    if (!f) atomicAdd(n_unfinished, 1);
  }
}
extern "C" int neurst_greedy_step(const float* logits, int32_t* preds, int32_t* finished, int32_t* len_decoded,
                                  int32_t* n_unfinished, int64_t B, int64_t V, int64_t step, int64_t max_steps, int32_t eos,
                                  void* stream) {
  NEURST_CHECK_ARG(B > 0 && V > 0 && step >= 0 && step < max_steps, "greedy_step: bad args");
  greedy_step_kernel<<<(int)B, 256, 0, as_stream(stream)>>>(logits, preds, finished, len_decoded, n_unfinished, V, step, max_steps, eos);
  NEURST_CHECK_LAUNCH("greedy_step");
  return NEURST_OK;
}
