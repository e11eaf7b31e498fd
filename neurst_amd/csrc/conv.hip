// a4: conv2d (SAME) + LayerNorm over channels + ReLU, and the 3x3 / stride-1 SAME max-pool of
// st/models/encoders/transformer_encoder.py:18-32,69-103 (tf.layers.conv2d, NHWC,
// kernel [kh,kw,Cin,Cout]).  Small-K convolution: one wavefront per output pixel with the
// output channels across the lanes, LN statistics by wavefront reduction, fused epilogue.
#include "conv.h"

#define LN_EPS 1e-6f
#define MAX_NCO 4  // Cout <= 256

static inline void same_pad(int64_t n, int k, int s, int64_t& out, int& before) {
  out = (n + s - 1) / s;
  int64_t total = (out - 1) * s + k - n;
  if (total < 0) total = 0;
  before = (int)(total / 2);
}

template <typename T>
__global__ void __launch_bounds__(256) conv_ln_relu_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               T* __restrict__ conv_out, T* __restrict__ out, float* __restrict__ mean,
                                                               float* __restrict__ rstd, ConvGeom g) {
  const int lane = threadIdx.x & 63;
  const int64_t npix = g.B * g.Ho * g.Wo;
  const int nco = (int)((g.Cout + 63) / 64);
  for (int64_t pix = blockIdx.x * 4 + (threadIdx.x >> 6); pix < npix; pix += (int64_t)gridDim.x * 4) {
    const int64_t wo = pix % g.Wo, r = pix / g.Wo, ho = r % g.Ho, b = r / g.Ho;
    float acc[MAX_NCO];
#pragma unroll
    for (int c = 0; c < MAX_NCO; ++c) { int co = lane + 64 * c; acc[c] = (c < nco && co < g.Cout) ? bias[co] : 0.f; }
    for (int i = 0; i < g.kh; ++i) {
      const int64_t hi = ho * g.sh + i - g.pt;
      if (hi < 0 || hi >= g.H) continue;
      for (int j = 0; j < g.kw; ++j) {
        const int64_t wi = wo * g.sw + j - g.pl;
        if (wi < 0 || wi >= g.W) continue;
        const T* xp = x + ((b * g.H + hi) * g.W + wi) * g.Cin;
        const float* wp = w + (int64_t)(i * g.kw + j) * g.Cin * g.Cout;
        for (int64_t ci = 0; ci < g.Cin; ++ci) {
          const float xv = to_f32<T>(xp[ci]);
#pragma unroll
          for (int c = 0; c < MAX_NCO; ++c) { int co = lane + 64 * c; if (c < nco && co < g.Cout) acc[c] += xv * wp[ci * g.Cout + co]; }
        }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAX_NCO; ++c) {
      int co = lane + 64 * c;
      if (c < nco && co < g.Cout) { acc[c] = to_f32<T>(from_f32<T>(acc[c])); conv_out[pix * g.Cout + co] = from_f32<T>(acc[c]); s += acc[c]; }
    }
    const float mu = wave_sum(s) / (float)g.Cout;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAX_NCO; ++c) { int co = lane + 64 * c; if (c < nco && co < g.Cout) { float d = acc[c] - mu; q += d * d; } }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)g.Cout + LN_EPS);
#pragma unroll
    for (int c = 0; c < MAX_NCO; ++c) {
      int co = lane + 64 * c;
      if (c < nco && co < g.Cout) out[pix * g.Cout + co] = from_f32<T>(fmaxf(((acc[c] - mu) * rs) * gamma[co] + beta[co], 0.f));
    }
    if (lane == 0) { mean[pix] = mu; rstd[pix] = rs; }
  }
}

// per pixel: ReLU gate + LN backward -> dconv; dgamma/dbeta/dbias partials per wave
template <typename T>
__global__ void __launch_bounds__(256) conv_ln_relu_bwd_pix_kernel(const T* __restrict__ dout, const T* __restrict__ conv_out, const T* __restrict__ out,
                                                                   const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, T* __restrict__ dconv, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, float* __restrict__ dbias, ConvGeom g) {
  const int lane = threadIdx.x & 63;
  const int64_t npix = g.B * g.Ho * g.Wo;
  const int nco = (int)((g.Cout + 63) / 64);
  float pg[MAX_NCO] = {0.f, 0.f, 0.f, 0.f}, pb[MAX_NCO] = {0.f, 0.f, 0.f, 0.f}, pc[MAX_NCO] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t pix = blockIdx.x * 4 + (threadIdx.x >> 6); pix < npix; pix += (int64_t)gridDim.x * 4) {
    const float mu = mean[pix], rs = rstd[pix];
    float gv[MAX_NCO], xh[MAX_NCO];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAX_NCO; ++c) {
      int co = lane + 64 * c;
      gv[c] = 0.f; xh[c] = 0.f;
      if (c < nco && co < g.Cout) {
        float d = to_f32<T>(out[pix * g.Cout + co]) > 0.f ? to_f32<T>(dout[pix * g.Cout + co]) : 0.f;
        xh[c] = (to_f32<T>(conv_out[pix * g.Cout + co]) - mu) * rs;
        pg[c] += d * xh[c]; pb[c] += d;
        gv[c] = d * gamma[co];
        s1 += gv[c]; s2 += gv[c] * xh[c];
      }
    }
    const float m1 = wave_sum(s1) / (float)g.Cout, m2 = wave_sum(s2) / (float)g.Cout;
#pragma unroll
    for (int c = 0; c < MAX_NCO; ++c) {
      int co = lane + 64 * c;
      if (c < nco && co < g.Cout) {
        float dc = rs * (gv[c] - m1 - xh[c] * m2);
        dc = to_f32<T>(from_f32<T>(dc));
        dconv[pix * g.Cout + co] = from_f32<T>(dc);
        pc[c] += dc;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < MAX_NCO; ++c) {
    int co = lane + 64 * c;
    if (c < nco && co < g.Cout) { atomicAdd(dgamma + co, pg[c]); atomicAdd(dbeta + co, pb[c]); atomicAdd(dbias + co, pc[c]); }
  }
}

// dw[i,j,ci,co] += sum_pix x[b, ho*sh+i-pt, wo*sw+j-pl, ci] * dconv[pix, co]; grid (kh*kw*Cin, nsplit)
template <typename T>
__global__ void __launch_bounds__(256) conv_dw_kernel(const T* __restrict__ x, const T* __restrict__ dconv, float* __restrict__ dw, ConvGeom g) {
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int tap = blockIdx.x;
  const int64_t ci = tap % g.Cin;
  const int ij = (int)(tap / g.Cin), i = ij / g.kw, j = ij % g.kw;
  const int64_t npix = g.B * g.Ho * g.Wo;
  const int64_t per = (npix + gridDim.y - 1) / gridDim.y;
  const int64_t p0 = per * blockIdx.y, p1 = p0 + per < npix ? p0 + per : npix;
  for (int64_t co0 = 0; co0 < g.Cout; co0 += 64) {
    const int64_t co = co0 + lane;
    float s = 0.f;
    for (int64_t pix = p0 + pl; pix < p1; pix += 4) {
      const int64_t wo = pix % g.Wo, r = pix / g.Wo, ho = r % g.Ho, b = r / g.Ho;
      const int64_t hi = ho * g.sh + i - g.pt, wi = wo * g.sw + j - g.pl;
      if (hi < 0 || hi >= g.H || wi < 0 || wi >= g.W) continue;
      const float xv = to_f32<T>(x[((b * g.H + hi) * g.W + wi) * g.Cin + ci]);
      if (co < g.Cout) s += xv * to_f32<T>(dconv[pix * g.Cout + co]);
    }
    __syncthreads();
    sm[pl][lane] = s;
    __syncthreads();
    if (pl == 0 && co < g.Cout) atomicAdd(dw + ((int64_t)ij * g.Cin + ci) * g.Cout + co, sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane]);
  }
}

// dx[b,h,w,ci] = sum_{i,j,co} dconv[b,ho,wo,co] * w[i,j,ci,co] over the taps that hit (h,w)
template <typename T>
__global__ void conv_dx_kernel(const T* __restrict__ dconv, const float* __restrict__ w, T* __restrict__ dx, ConvGeom g) {
  const int64_t total = g.B * g.H * g.W * g.Cin;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t ci = idx % g.Cin, r = idx / g.Cin, wi = r % g.W, r2 = r / g.W, hi = r2 % g.H, b = r2 / g.H;
    float s = 0.f;
    for (int i = 0; i < g.kh; ++i) {
      const int64_t hn = hi + g.pt - i;
      if (hn < 0 || hn % g.sh) continue;
      const int64_t ho = hn / g.sh;
      if (ho >= g.Ho) continue;
      for (int j = 0; j < g.kw; ++j) {
        const int64_t wn = wi + g.pl - j;
        if (wn < 0 || wn % g.sw) continue;
        const int64_t wo = wn / g.sw;
        if (wo >= g.Wo) continue;
        const T* dp = dconv + ((b * g.Ho + ho) * g.Wo + wo) * g.Cout;
        const float* wp = w + ((int64_t)(i * g.kw + j) * g.Cin + ci) * g.Cout;
        for (int64_t co = 0; co < g.Cout; ++co) s += to_f32<T>(dp[co]) * wp[co];
      }
    }
    dx[idx] = from_f32<T>(s);
  }
}

static int make_geom(ConvGeom& g, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kh, int kw, int sh, int sw) {
  NEURST_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cout <= 64 * MAX_NCO && kh > 0 && kw > 0 && sh > 0 && sw > 0,
                   "conv2d: bad shape (Cout must be <= %d)", 64 * MAX_NCO);
  g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.kh = kh; g.kw = kw; g.sh = sh; g.sw = sw;
  same_pad(H, kh, sh, g.Ho, g.pt);
  same_pad(W, kw, sw, g.Wo, g.pl);
  return NEURST_OK;
}
static inline int pix_grid(int64_t npix) { int64_t x = cdiv64(npix, 4); return (int)(x > 4096 ? 4096 : (x < 1 ? 1 : x)); }

int g_opt_conv_fast = 1;                     // neurst_set_option("conv_fast", 0): general kernels only (A/B tests)

extern "C" int64_t neurst_conv2d_bwd_workspace_bytes(int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kh, int kw, int sh,
                                                     int sw, int dtype) {
  ConvGeom g;
  if (dtype != NEURST_BF16 || make_geom(g, B, H, W, Cin, Cout, kh, kw, sh, sw) != NEURST_OK) return 0;
  if (kh * kw * Cin <= 32 && Cout <= 64) return 0;      // first-layer kernel: no im2col image
  return neurst_conv_col_bytes(g);
}

extern "C" int neurst_conv2d_ln_relu_fwd(const void* x, const float* w, const float* bias, const float* gamma, const float* beta,
                                         void* conv_out, void* out, float* mean, float* rstd, int64_t B, int64_t H, int64_t W,
                                         int64_t Cin, int64_t Cout, int kh, int kw, int sh, int sw, int dtype, void* stream) {
  ConvGeom g;
  int rc = make_geom(g, B, H, W, Cin, Cout, kh, kw, sh, sw);
  if (rc) return rc;
  int grid = pix_grid(g.B * g.Ho * g.Wo);
  if (dtype == NEURST_BF16 && g_opt_conv_fast &&
      neurst_conv_fwd_fast((const bf16_t*)x, w, bias, gamma, beta, (bf16_t*)conv_out, (bf16_t*)out, mean, rstd, g, as_stream(stream))) {
    NEURST_CHECK_LAUNCH("conv2d_ln_relu_fwd(fast)");
    return NEURST_OK;
  }
  if (dtype == NEURST_BF16)
    conv_ln_relu_fwd_kernel<bf16_t><<<grid, 256, 0, as_stream(stream)>>>((const bf16_t*)x, w, bias, gamma, beta, (bf16_t*)conv_out, (bf16_t*)out, mean, rstd, g);
  else
    conv_ln_relu_fwd_kernel<float><<<grid, 256, 0, as_stream(stream)>>>((const float*)x, w, bias, gamma, beta, (float*)conv_out, (float*)out, mean, rstd, g);
  NEURST_CHECK_LAUNCH("conv2d_ln_relu_fwd");
  return NEURST_OK;
}

extern "C" int neurst_conv2d_ln_relu_bwd(const void* dout, const void* x, const float* w, const float* gamma, const void* conv_out,
                                         const void* out, const float* mean, const float* rstd, void* dconv_ws, void* col_ws,
                                         int64_t col_ws_bytes, void* dx, float* dw, float* dbias, float* dgamma, float* dbeta, int64_t B,
                                         int64_t H, int64_t W, int64_t Cin, int64_t Cout, int kh, int kw, int sh, int sw, int dtype,
                                         void* stream) {
  ConvGeom g;
  int rc = make_geom(g, B, H, W, Cin, Cout, kh, kw, sh, sw);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  const int64_t npix = g.B * g.Ho * g.Wo;
  if (dtype == NEURST_BF16 && g_opt_conv_fast) {
    // every step falls back to its general kernel on its own when the shape is not one the fast path handles
    const bf16_t* xb = (const bf16_t*)x;
    bf16_t* dc = (bf16_t*)dconv_ws;
    int grid1 = pix_grid(npix);
    if (grid1 > 1024) grid1 = 1024;
    if (!neurst_conv_bwd_pix_fast((const bf16_t*)dout, (const bf16_t*)conv_out, (const bf16_t*)out, gamma, mean, rstd, dc, dgamma, dbeta, dbias, g, s))
      conv_ln_relu_bwd_pix_kernel<bf16_t><<<grid1, 256, 0, s>>>((const bf16_t*)dout, (const bf16_t*)conv_out, (const bf16_t*)out, gamma, mean, rstd, dc, dgamma, dbeta, dbias, g);
    if (!neurst_conv_dw_fast(xb, dc, dw, col_ws, col_ws_bytes, g, s)) {
      int64_t ns = cdiv64(npix, 2048);
      if (ns > 64) ns = 64;
      conv_dw_kernel<bf16_t><<<dim3((unsigned)(kh * kw * Cin), (unsigned)ns), 256, 0, s>>>(xb, dc, dw, g);
    }
    if (dx && !neurst_conv_dx_fast(dc, w, (bf16_t*)dx, g, s)) {
      const int64_t nd = B * H * W * Cin;
      conv_dx_kernel<bf16_t><<<(int)(cdiv64(nd, 256) > 8192 ? 8192 : cdiv64(nd, 256)), 256, 0, s>>>(dc, w, (bf16_t*)dx, g);
    }
    NEURST_CHECK_LAUNCH("conv2d_ln_relu_bwd(fast)");
    return NEURST_OK;
  }
  int grid = pix_grid(npix);
  if (grid > 1024) grid = 1024;
  int64_t nsplit = cdiv64(npix, 2048);
  if (nsplit > 64) nsplit = 64;
  dim3 gdw((unsigned)(kh * kw * Cin), (unsigned)nsplit);
  int64_t ndx = B * H * W * Cin;
  int gdx = (int)(cdiv64(ndx, 256) > 8192 ? 8192 : cdiv64(ndx, 256));
#define RUN(TT)                                                                                                                            \
  conv_ln_relu_bwd_pix_kernel<TT><<<grid, 256, 0, s>>>((const TT*)dout, (const TT*)conv_out, (const TT*)out, gamma, mean, rstd, (TT*)dconv_ws, \
                                                       dgamma, dbeta, dbias, g);                                                           \
  conv_dw_kernel<TT><<<gdw, 256, 0, s>>>((const TT*)x, (const TT*)dconv_ws, dw, g);                                                        \
  if (dx) conv_dx_kernel<TT><<<gdx, 256, 0, s>>>((const TT*)dconv_ws, w, (TT*)dx, g);
  if (dtype == NEURST_BF16) { RUN(bf16_t) } else { RUN(float) }
#undef RUN
  NEURST_CHECK_LAUNCH("conv2d_ln_relu_bwd");
  return NEURST_OK;
}

// ---------------------------------------------------------------- 3x3 stride-1 SAME max pool
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, int8_t* __restrict__ arg, int64_t B, int64_t H, int64_t W, int64_t C) {
  const int64_t total = B * H * W * C;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = idx % C, r = idx / C, wi = r % W, r2 = r / W, hi = r2 % H, b = r2 / H;
    float best = -INFINITY;
    int bk = 4;
    for (int k = 0; k < 9; ++k) {
      const int64_t h2 = hi + k / 3 - 1, w2 = wi + k % 3 - 1;
      if (h2 < 0 || h2 >= H || w2 < 0 || w2 >= W) continue;
      float v = to_f32<T>(x[((b * H + h2) * W + w2) * C + c]);
      if (v > best) { best = v; bk = k; }
    }
    out[idx] = from_f32<T>(best);
    arg[idx] = (int8_t)bk;
  }
}
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dout, const int8_t* __restrict__ arg, T* __restrict__ dx, int64_t B, int64_t H, int64_t W, int64_t C) {
  const int64_t total = B * H * W * C;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = idx % C, r = idx / C, wi = r % W, r2 = r / W, hi = r2 % H, b = r2 / H;
    float s = 0.f;
    for (int k = 0; k < 9; ++k) {   // output (h2,w2) picked this input iff its argmax tap points back here
      const int64_t h2 = hi - (k / 3 - 1), w2 = wi - (k % 3 - 1);
      if (h2 < 0 || h2 >= H || w2 < 0 || w2 >= W) continue;
      const int64_t o = ((b * H + h2) * W + w2) * C + c;
      if (arg[o] == k) s += to_f32<T>(dout[o]);
    }
    dx[idx] = from_f32<T>(s);
  }
}
// bf16, C % 8 == 0: 8 channels (16 bytes) per thread
__global__ void __launch_bounds__(256) maxpool_fwd_vec_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int8_t* __restrict__ arg,
                                                              int64_t B, int64_t H, int64_t W, int64_t C) {
  const int64_t c8n = C / 8, total = B * H * W * c8n;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c8 = idx % c8n, r = idx / c8n, wi = r % W, r2 = r / W, hi = r2 % H, b = r2 / H;
    float best[8];
    int bk[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bk[e] = 4; }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int64_t h2 = hi + k / 3 - 1, w2 = wi + k % 3 - 1;
      if (h2 < 0 || h2 >= H || w2 < 0 || w2 >= W) continue;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((b * H + h2) * W + w2) * C + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if ((float)v[e] > best[e]) { best[e] = (float)v[e]; bk[e] = k; }
    }
    bf16x8 o;
    unsigned lo = 0, hi4 = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)best[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) { lo |= (unsigned)bk[e] << (8 * e); hi4 |= (unsigned)bk[4 + e] << (8 * e); }
    *reinterpret_cast<bf16x8*>(out + r * C + c8 * 8) = o;
    *reinterpret_cast<u32x2*>(arg + r * C + c8 * 8) = u32x2{lo, hi4};
  }
}
__global__ void __launch_bounds__(256) maxpool_bwd_vec_kernel(const bf16_t* __restrict__ dout, const int8_t* __restrict__ arg,
                                                              bf16_t* __restrict__ dx, int64_t B, int64_t H, int64_t W, int64_t C) {
  const int64_t c8n = C / 8, total = B * H * W * c8n;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c8 = idx % c8n, r = idx / c8n, wi = r % W, r2 = r / W, hi = r2 % H, b = r2 / H;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {   // output (h2,w2) picked this input iff its argmax tap points back here
      const int64_t h2 = hi - (k / 3 - 1), w2 = wi - (k % 3 - 1);
      if (h2 < 0 || h2 >= H || w2 < 0 || w2 >= W) continue;
      const int64_t o = ((b * H + h2) * W + w2) * C + c8 * 8;
      const u32x2 a = *reinterpret_cast<const u32x2*>(arg + o);
      const bf16x8 d = *reinterpret_cast<const bf16x8*>(dout + o);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if ((int)(((e < 4 ? a[0] : a[1]) >> (8 * (e & 3))) & 0xff) == k) s[e] += (float)d[e];
    }
    bf16x8 o8;
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = (bf16_t)s[e];
    *reinterpret_cast<bf16x8*>(dx + r * C + c8 * 8) = o8;
  }
}
static inline bool pool_vec_ok(const void* a, const void* b, const void* c, int64_t C) {
  return C % 8 == 0 && ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0 && (((uintptr_t)c) & 7) == 0;
}
extern "C" int neurst_maxpool3x3_fwd(const void* x, void* out, int8_t* argidx, int64_t B, int64_t H, int64_t W, int64_t C, int dtype, void* stream) {
  NEURST_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0, "maxpool: bad shape");
  int64_t n = B * H * W * C;
  int grid = (int)(cdiv64(n, 256) > 8192 ? 8192 : cdiv64(n, 256));
  if (dtype == NEURST_BF16 && pool_vec_ok(x, out, argidx, C)) {
    const int64_t n8 = n / 8;
    maxpool_fwd_vec_kernel<<<(int)(cdiv64(n8, 256) > 8192 ? 8192 : cdiv64(n8, 256)), 256, 0, as_stream(stream)>>>((const bf16_t*)x, (bf16_t*)out, argidx, B, H, W, C);
  } else if (dtype == NEURST_BF16) maxpool_fwd_kernel<bf16_t><<<grid, 256, 0, as_stream(stream)>>>((const bf16_t*)x, (bf16_t*)out, argidx, B, H, W, C);
  else maxpool_fwd_kernel<float><<<grid, 256, 0, as_stream(stream)>>>((const float*)x, (float*)out, argidx, B, H, W, C);
  NEURST_CHECK_LAUNCH("maxpool3x3_fwd");
  return NEURST_OK;
}
extern "C" int neurst_maxpool3x3_bwd(const void* dout, const int8_t* argidx, void* dx, int64_t B, int64_t H, int64_t W, int64_t C, int dtype, void* stream) {
  NEURST_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0, "maxpool: bad shape");
  int64_t n = B * H * W * C;
  int grid = (int)(cdiv64(n, 256) > 8192 ? 8192 : cdiv64(n, 256));
  if (dtype == NEURST_BF16 && pool_vec_ok(dout, dx, argidx, C)) {
    const int64_t n8 = n / 8;
    maxpool_bwd_vec_kernel<<<(int)(cdiv64(n8, 256) > 8192 ? 8192 : cdiv64(n8, 256)), 256, 0, as_stream(stream)>>>((const bf16_t*)dout, argidx, (bf16_t*)dx, B, H, W, C);
  } else if (dtype == NEURST_BF16) maxpool_bwd_kernel<bf16_t><<<grid, 256, 0, as_stream(stream)>>>((const bf16_t*)dout, argidx, (bf16_t*)dx, B, H, W, C);
  else maxpool_bwd_kernel<float><<<grid, 256, 0, as_stream(stream)>>>((const float*)dout, argidx, (float*)dx, B, H, W, C);
  NEURST_CHECK_LAUNCH("maxpool3x3_bwd");
  return NEURST_OK;
}
