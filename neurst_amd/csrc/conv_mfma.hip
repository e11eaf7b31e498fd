// a4, bf16 fast paths of the conv + LayerNorm + ReLU front end (st/models/encoders/transformer_encoder.py:18-32,69-95:
// tf.layers.conv2d SAME, NHWC, kernel [kh,kw,Cin,Cout], layer_norm over the channels, ReLU).  conv.hip keeps the general
// kernels (any shape, fp32 parity mode) and the C ABI; it asks the functions at the end of this file first.
//
//  * 3 x 3 layers with Cin % 32 == 0 (every layer after the first): implicit GEMM on the matrix cores.  Rows = output pixels,
//    columns = output channels, reduction = (tap, input channel).  The A operand of v_mfma_f32_16x16x32_bf16 wants, per lane,
//    8 consecutive reduction elements of one row - 16 contiguous bytes of the NHWC input at (pixel + tap, ci .. ci + 7) - so
//    A fragments come STRAIGHT from global memory (zeros where the tap leaves the image; neighbouring taps hit the caches),
//    18 loads in flight per wave at Cin = 64.  The whole weight tensor sits in LDS as a bf16 [Cout][9 Cin] image, staged once
//    per (persistent) workgroup.  Epilogue: bias, rounding to bf16, LayerNorm statistics over the channels of a pixel (4
//    lanes per pixel after a pass through a per-wave LDS tile), ReLU, row-contiguous stores of conv_out and out.
//    The data gradient dX is the same kernel with the roles swapped: rows = INPUT pixels, columns = input channels,
//    reduction = (tap, output channel), A fragments from dconv at the output pixel a tap maps to (if any).
//    HBM-bound by design: conv_1 of the LUT front end moves 185 MB forward for 17.7 GFLOP.
//  * the first layer (Cin = 1 .. 3, at most 32 reduction elements): no matrix shape fits; one wave per pixel with the output
//    channel on the lane as in the general kernel, but the weights live in registers (the general kernel re-reads them for
//    every pixel: 2-7 KB of L1 traffic per 128-byte output row), the input patch is ONE vector load whose lanes are broadcast
//    with v_readlane, and waves are persistent.  Its weight gradient is the same loop with 27 accumulators per lane.
//  * weight gradient of the 3 x 3 / Cin % 8 == 0 layers: im2col image [pixels][9 Cin] in a caller workspace + the TN GEMM of
//    gemm.hip (reduction over the pixels, split-K) - 276 MB written and read once at conv_1's size.
//  * ReLU gate + LayerNorm backward per pixel with 16-byte lanes (Cout / 8 lanes per pixel) and ONE set of atomics per
//    workgroup for dgamma / dbeta / dbias.
#include "conv.h"

#define LN_EPS 1e-6f

extern "C" int neurst_gemm_bias_act(const void* A, const void* B, void* C, const float* bias, const void* relu_src, int64_t M,
                                    int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int trans_a, int trans_b,
                                    int ab_dtype, int c_dtype, int relu, int accumulate, int split_k, void* stream);

namespace cv {

constexpr int NTH = 512, NWV = 8, PADK = 8;

template <int NJ, int NC, bool DX, bool GATHER = false>
constexpr size_t mfma_lds_bytes() {
  return (size_t)(16 * NJ) * ((GATHER ? 32 : 9 * 32 * NC) + PADK) * 2 + (size_t)NWV * 16 * (16 * NJ + 4) * 4 + 3 * 16 * NJ * 4;
}

// NJ = 16-column MFMA tiles of the output (N = 16 NJ channels), NC = 32-element chunks of the reduction channels (C = 32 NC)
// GATHER (forward only): a layer with at most 32 reduction elements in all (kh kw Cin <= 32: the first layer, Cin = 1 .. 3) as
// ONE K-step - every lane gathers its 8 reduction elements (tap, ci) with 2-byte loads, the weight image is zero beyond K.
template <int NJ, int NC, bool DX, bool GATHER = false>
__global__ void __launch_bounds__(NTH) conv3x3_mfma_kernel(const bf16_t* __restrict__ src, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, bf16_t* __restrict__ dst0,
                                                           bf16_t* __restrict__ dst1, float* __restrict__ mean, float* __restrict__ rstd,
                                                           ConvGeom g) {
  constexpr int N = 16 * NJ, C = 32 * NC, NS = GATHER ? 1 : 9 * NC, K = 32 * NS, LDW = K + PADK, LDC = N + 4, CW = 4 * NJ;
  static_assert(!(GATHER && DX), "the gathering variant is a forward kernel");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* const wl = reinterpret_cast<bf16_t*>(smem);                                 // [N][LDW]: row n, reduction index contiguous
  float* const stage = reinterpret_cast<float*>(smem + (size_t)N * LDW * 2);          // [NWV][16][LDC]
  float* const par = stage + NWV * 16 * LDC;                                          // bias, gamma, beta [3][N] (forward)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Cin = (int)g.Cin, Cout = (int)g.Cout;
  const int Kt = g.kh * g.kw * Cin;                                                   // reduction elements in all
  if constexpr (GATHER) {
    for (int idx = tid; idx < N * LDW; idx += NTH) wl[idx] = (bf16_t)0.f;
    __syncthreads();
    for (int idx = tid; idx < Kt * Cout; idx += NTH) wl[(idx % Cout) * LDW + idx / Cout] = (bf16_t)w[idx];   // [k][co] -> [co][k]
  } else {
    for (int idx = tid; idx < 9 * Cin * Cout; idx += NTH) {                          // global layout [tap][ci][co]
      const int co = idx % Cout, r = idx / Cout, ci = r % Cin, tap = r / Cin;
      const bf16_t v = (bf16_t)w[idx];
      if constexpr (!DX) wl[co * LDW + tap * C + ci] = v;
      else wl[ci * LDW + tap * C + co] = v;
    }
  }
  if constexpr (!DX) {
    for (int i = tid; i < N; i += NTH) { par[i] = bias[i]; par[N + i] = gamma[i]; par[2 * N + i] = beta[i]; }
  }
  __syncthreads();
  const int64_t M = DX ? g.B * g.H * g.W : g.B * g.Ho * g.Wo;
  const int64_t ntiles = (M + 16 * NWV - 1) / (16 * NWV);
  const int r16 = lane & 15, kg = lane >> 4;
  float* const Cs = stage + wave * 16 * LDC;
  const bf16x8 zero8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
  // GATHER: this lane's reduction elements k = 8 kg + e as (tap row i, tap column j, offset from the pixel's first input element)
  int gi[8], gj[8], goff[8];
  if constexpr (GATHER) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = kg * 8 + e, tap = k / Cin, ci = k - tap * Cin;
      gi[e] = k < Kt ? tap / g.kw : -(1 << 20);                                    // beyond K: never inside the image
      gj[e] = tap % g.kw;
      goff[e] = ((tap / g.kw) * (int)g.W + gj[e]) * Cin + ci;
    }
  }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t p = tile * (16 * NWV) + wave * 16 + r16;
    const bool pv = p < M;
    bf16x8 a[NS];
    if constexpr (GATHER) {
      const int64_t wo = p % g.Wo, t = p / g.Wo, ho = t % g.Ho, b = t / g.Ho;
      const int h0 = (int)ho * g.sh - g.pt, w0 = (int)wo * g.sw - g.pl;
      const int64_t base = ((b * g.H + h0) * g.W + w0) * Cin;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool ok = pv && (unsigned)(h0 + gi[e]) < (unsigned)g.H && (unsigned)(w0 + gj[e]) < (unsigned)g.W;
        a[0][e] = ok ? src[base + goff[e]] : (bf16_t)0.f;
      }
    } else if constexpr (!DX) {
      const int64_t wo = p % g.Wo, t = p / g.Wo, ho = t % g.Ho, b = t / g.Ho;
      const int h0 = (int)ho * g.sh - g.pt, w0 = (int)wo * g.sw - g.pl;
      const int64_t base = ((b * g.H + h0) * g.W + w0) * Cin + kg * 8;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const bool ok = pv && (unsigned)(h0 + i) < (unsigned)g.H && (unsigned)(w0 + j) < (unsigned)g.W;
#pragma unroll
          for (int c = 0; c < NC; ++c)
            a[(i * 3 + j) * NC + c] = ok ? *reinterpret_cast<const bf16x8*>(src + base + ((int64_t)i * g.W + j) * Cin + c * 32) : zero8;
        }
    } else {
      const int64_t wi = p % g.W, t = p / g.W, hi = t % g.H, b = t / g.H;
      int hoi[3], woj[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int hn = (int)hi + g.pt - i, wn = (int)wi + g.pl - i;
        hoi[i] = (hn >= 0 && hn % g.sh == 0 && hn / g.sh < (int)g.Ho) ? hn / g.sh : -1;
        woj[i] = (wn >= 0 && wn % g.sw == 0 && wn / g.sw < (int)g.Wo) ? wn / g.sw : -1;
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const bool ok = pv && hoi[i] >= 0 && woj[j] >= 0;
          const int64_t off = ((b * g.Ho + hoi[i]) * g.Wo + woj[j]) * Cout + kg * 8;
#pragma unroll
          for (int c = 0; c < NC; ++c) a[(i * 3 + j) * NC + c] = ok ? *reinterpret_cast<const bf16x8*>(src + off + c * 32) : zero8;
        }
    }
    f32x4 acc[NJ];
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) acc[jn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int jn = 0; jn < NJ; ++jn) {
        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(wl + (jn * 16 + r16) * LDW + s * 32 + kg * 8);
        acc[jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s], bfr, acc[jn], 0, 0, 0);
      }
    // accumulators (row = 4 kg + r, column = 16 jn + r16) -> this wave's LDS tile -> 4 lanes per pixel row
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cs[(kg * 4 + r) * LDC + jn * 16 + r16] = acc[jn][r] + (DX ? 0.f : par[jn * 16 + r16]);
    // the same wave reads back what it wrote: LDS operations of one wave complete in order
    const int row = lane >> 2, cq = lane & 3;
    const int64_t pr = tile * (16 * NWV) + wave * 16 + row;
    float v[CW];
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(Cs + row * LDC + cq * CW + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * q + e] = (float)(bf16_t)t4[e];              // rounded as stored
    }
    if constexpr (DX) {
      if (pr < M) {
#pragma unroll
        for (int q = 0; q < NJ; ++q)
          *reinterpret_cast<bf16x4*>(dst0 + pr * N + cq * CW + 4 * q) = bf16x4{(bf16_t)v[4 * q], (bf16_t)v[4 * q + 1], (bf16_t)v[4 * q + 2], (bf16_t)v[4 * q + 3]};
      }
    } else {
      float s1 = 0.f;
#pragma unroll
      for (int e = 0; e < CW; ++e) s1 += v[e];
      s1 += __shfl_xor(s1, 1, 64);
      s1 += __shfl_xor(s1, 2, 64);
      const float mu = s1 / (float)N;
      float q2 = 0.f;
#pragma unroll
      for (int e = 0; e < CW; ++e) { const float d = v[e] - mu; q2 += d * d; }
      q2 += __shfl_xor(q2, 1, 64);
      q2 += __shfl_xor(q2, 2, 64);
      const float rs = 1.0f / sqrtf(q2 / (float)N + LN_EPS);
      if (pr < M) {
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
          const int c0 = cq * CW + 4 * q;
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = fmaxf(((v[4 * q + e] - mu) * rs) * par[N + c0 + e] + par[2 * N + c0 + e], 0.f);
          *reinterpret_cast<bf16x4*>(dst0 + pr * N + c0) = bf16x4{(bf16_t)v[4 * q], (bf16_t)v[4 * q + 1], (bf16_t)v[4 * q + 2], (bf16_t)v[4 * q + 3]};
          *reinterpret_cast<bf16x4*>(dst1 + pr * N + c0) = bf16x4{(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
        }
        if (cq == 0) { mean[pr] = mu; rstd[pr] = rs; }
      }
    }
  }
}

__device__ __forceinline__ float lane_bcast(float v, int k) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}

// (b, ho, wo) of an output pixel; 32-bit division when the pixel count allows (wave-uniform operands: scalar unit)
struct PixPos { int b, ho, wo; };
__device__ __forceinline__ PixPos pix_pos(const ConvGeom& g, int64_t pix, bool small) {
  PixPos r;
  if (small) {
    const unsigned p = (unsigned)pix, Wo = (unsigned)g.Wo, Ho = (unsigned)g.Ho;
    const unsigned t = p / Wo;
    r.wo = (int)(p - t * Wo);
    r.b = (int)(t / Ho);
    r.ho = (int)(t - (unsigned)r.b * Ho);
  } else {
    const int64_t t = pix / g.Wo;
    r.wo = (int)(pix % g.Wo);
    r.ho = (int)(t % g.Ho);
    r.b = (int)(t / g.Ho);
  }
  return r;
}
__device__ __forceinline__ void pix_next(const ConvGeom& g, PixPos& r) {
  if (++r.wo == (int)g.Wo) { r.wo = 0; if (++r.ho == (int)g.Ho) { r.ho = 0; ++r.b; } }
}
// input element that reduction index `lane` (= tap * Cin + ci) of the output pixel at `r` reads, or 0 outside the image / beyond K
__device__ __forceinline__ float patch_elem(const bf16_t* __restrict__ x, const ConvGeom& g, const PixPos& r, bool kvalid, int ti, int tj, int tci) {
  const int hi = r.ho * g.sh + ti - g.pt, wi = r.wo * g.sw + tj - g.pl;
  const bool ok = kvalid && hi >= 0 && hi < (int)g.H && wi >= 0 && wi < (int)g.W;
  return ok ? (float)x[(((int64_t)r.b * g.H + hi) * g.W + wi) * g.Cin + tci] : 0.f;
}

// first layer: K = kh kw Cin <= KM <= 32 reduction elements, Cout <= 64; lane = output channel, weights in registers
template <int KM>
__global__ void __launch_bounds__(256) conv_smallk_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, bf16_t* __restrict__ conv_out,
                                                              bf16_t* __restrict__ out, float* __restrict__ mean, float* __restrict__ rstd,
                                                              ConvGeom g, int K) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Cout = (int)g.Cout;
  const bool cvd = lane < Cout;
  float wr[KM];
#pragma unroll
  for (int k = 0; k < KM; ++k) wr[k] = (k < K && cvd) ? w[k * Cout + lane] : 0.f;
  const float bs = cvd ? bias[lane] : 0.f, gm = cvd ? gamma[lane] : 0.f, bt = cvd ? beta[lane] : 0.f;
  const bool kvalid = lane < K;
  const int tap = kvalid ? lane / (int)g.Cin : 0, tci = kvalid ? lane % (int)g.Cin : 0, ti = tap / g.kw, tj = tap % g.kw;
  const int64_t npix = g.B * g.Ho * g.Wo;
  const float invC = 1.f / (float)Cout;
  constexpr int PP = 4;
  __shared__ __attribute__((aligned(16))) bf16_t st[4][2][PP][64];   // per wave: 4 pixel rows of conv_out and of out
  const bool small = npix < (1ll << 31);
  for (int64_t p0 = ((int64_t)blockIdx.x * 4 + wv) * PP; p0 < npix; p0 += (int64_t)gridDim.x * 4 * PP) {
    float xv[PP];
    PixPos pos = pix_pos(g, p0, small);
#pragma unroll
    for (int q = 0; q < PP; ++q) {
      xv[q] = (p0 + q < npix) ? patch_elem(x, g, pos, kvalid, ti, tj, tci) : 0.f;
      pix_next(g, pos);
    }
    const bool packed = Cout == 64 && p0 + PP <= npix;       // 4 whole pixel rows = 512 contiguous bytes per tensor
#pragma unroll
    for (int q = 0; q < PP; ++q) {
      const int64_t pix = p0 + q;
      if (pix >= npix) break;
      float acc = bs;
#pragma unroll
      for (int k = 0; k < KM; ++k) acc = fmaf(lane_bcast(xv[q], k), wr[k], acc);
      acc = (float)(bf16_t)acc;
      const float mu = wave_sum_dpp(cvd ? acc : 0.f) * invC;
      const float d = cvd ? acc - mu : 0.f;
      const float rs = 1.0f / sqrtf(wave_sum_dpp(d * d) * invC + LN_EPS);
      const bf16_t o = (bf16_t)fmaxf((d * rs) * gm + bt, 0.f);
      if (packed) {
        st[wv][0][q][lane] = (bf16_t)acc;
        st[wv][1][q][lane] = o;
      } else if (cvd) {
        conv_out[pix * Cout + lane] = (bf16_t)acc;
        out[pix * Cout + lane] = o;
      }
      if (lane == 0) { mean[pix] = mu; rstd[pix] = rs; }
    }
    if (packed) {   // the same wave reads back what it wrote (LDS operations of one wave complete in order): 8 bytes per lane
      const u32x2 c2 = *reinterpret_cast<const u32x2*>(&st[wv][0][0][0] + lane * 4);
      const u32x2 o2 = *reinterpret_cast<const u32x2*>(&st[wv][1][0][0] + lane * 4);
      *reinterpret_cast<u32x2*>(conv_out + p0 * 64 + lane * 4) = c2;
      *reinterpret_cast<u32x2*>(out + p0 * 64 + lane * 4) = o2;
    }
  }
}

// weight gradient of the first layer: dw[k][co] += sum_pix patch[pix][k] * dconv[pix][co]; lane = output channel, KM accumulators
template <int KM>
__global__ void __launch_bounds__(1024) conv_smallk_dw_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dconv,
                                                             float* __restrict__ dw, ConvGeom g, int K) {
  constexpr int NWD = 16;                                      // waves per workgroup: one set of atomics per 16 waves
  extern __shared__ __attribute__((aligned(16))) char smem_dw[];
  float (*red)[KM][64] = reinterpret_cast<float (*)[KM][64]>(smem_dw);   // [NWD][KM][64]
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Cout = (int)g.Cout;
  const bool cvd = lane < Cout;
  const bool kvalid = lane < K;
  const int tap = kvalid ? lane / (int)g.Cin : 0, tci = kvalid ? lane % (int)g.Cin : 0, ti = tap / g.kw, tj = tap % g.kw;
  const int64_t npix = g.B * g.Ho * g.Wo;
  float acc[KM];
#pragma unroll
  for (int k = 0; k < KM; ++k) acc[k] = 0.f;
  constexpr int PP = 8;
  const bool small = npix < (1ll << 31);
  for (int64_t p0 = ((int64_t)blockIdx.x * NWD + wv) * PP; p0 < npix; p0 += (int64_t)gridDim.x * NWD * PP) {
    float xv[PP], dv[PP];
    PixPos pos = pix_pos(g, p0, small);
#pragma unroll
    for (int q = 0; q < PP; ++q) {
      const bool in = p0 + q < npix;
      xv[q] = in ? patch_elem(x, g, pos, kvalid, ti, tj, tci) : 0.f;
      dv[q] = (in && cvd) ? (float)dconv[(p0 + q) * Cout + lane] : 0.f;
      pix_next(g, pos);
    }
#pragma unroll
    for (int q = 0; q < PP; ++q)
#pragma unroll
      for (int k = 0; k < KM; ++k) acc[k] = fmaf(lane_bcast(xv[q], k), dv[q], acc[k]);
  }
#pragma unroll
  for (int k = 0; k < KM; ++k) red[wv][k][lane] = acc[k];
  __syncthreads();
  for (int i = threadIdx.x; i < K * 64; i += 64 * NWD) {
    const int k = i >> 6, co = i & 63;
    float t = 0.f;
#pragma unroll
    for (int v = 0; v < NWD; ++v) t += red[v][k][co];
    if (co < Cout) atomicAdd(dw + (int64_t)k * Cout + co, t);
  }
}

// ReLU gate + LayerNorm backward per pixel -> dconv, and the column sums dgamma / dbeta / dbias; LPP = Cout / 8 lanes per pixel
template <int LPP>
__global__ void __launch_bounds__(256) conv_bwd_pix_vec_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ conv_out,
                                                               const bf16_t* __restrict__ out, const float* __restrict__ gamma,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               bf16_t* __restrict__ dconv, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ dbias, int64_t npix) {
  constexpr int Cout = 8 * LPP, PPW = 64 / LPP;
  __shared__ float red[4][3][Cout];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int sub = lane % LPP, pl = lane / LPP;
  float gm[8], pg[8], pb[8], pc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { gm[e] = gamma[sub * 8 + e]; pg[e] = 0.f; pb[e] = 0.f; pc[e] = 0.f; }
  const float invC = 1.f / (float)Cout;
  for (int64_t p0 = ((int64_t)blockIdx.x * 4 + wv) * PPW; p0 < npix; p0 += (int64_t)gridDim.x * 4 * PPW) {
    const int64_t pix = p0 + pl;
    const bool in = pix < npix;
    float gv[8], xh[8];
    float s1 = 0.f, s2 = 0.f, rs = 0.f;
    if (in) {
      const int64_t off = pix * Cout + sub * 8;
      const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dout + off);
      const bf16x8 cvv = *reinterpret_cast<const bf16x8*>(conv_out + off);
      const bf16x8 ov = *reinterpret_cast<const bf16x8*>(out + off);
      const float mu = mean[pix];
      rs = rstd[pix];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)ov[e] > 0.f ? (float)dv[e] : 0.f;
        xh[e] = ((float)cvv[e] - mu) * rs;
        pg[e] += d * xh[e];
        pb[e] += d;
        gv[e] = d * gm[e];
        s1 += gv[e];
        s2 += gv[e] * xh[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { gv[e] = 0.f; xh[e] = 0.f; }
    }
#pragma unroll
    for (int o = 1; o < LPP; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    const float m1 = s1 * invC, m2 = s2 * invC;
    if (in) {
      bf16x8 o8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o8[e] = (bf16_t)(rs * (gv[e] - m1 - xh[e] * m2));
        pc[e] += (float)o8[e];
      }
      *reinterpret_cast<bf16x8*>(dconv + pix * Cout + sub * 8) = o8;
    }
  }
  // lanes with the same channel group: sum over the pixels of the wave, then over the waves, one set of atomics per workgroup
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int o = LPP; o < 64; o <<= 1) { pg[e] += __shfl_xor(pg[e], o, 64); pb[e] += __shfl_xor(pb[e], o, 64); pc[e] += __shfl_xor(pc[e], o, 64); }
  if (lane < LPP) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[wv][0][sub * 8 + e] = pg[e]; red[wv][1][sub * 8 + e] = pb[e]; red[wv][2][sub * 8 + e] = pc[e]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * Cout; i += 256) {
    const int which = i / Cout, c = i % Cout;
    const float s = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    atomicAdd((which == 0 ? dgamma : which == 1 ? dbeta : dbias) + c, s);
  }
}

// im2col of a 3 x 3 layer: col[pix][tap * Cin + ci], 16 bytes per thread
__global__ void __launch_bounds__(256) im2col3x3_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, ConvGeom g) {
  const int c8n = (int)g.Cin / 8;
  const int64_t total = g.B * g.Ho * g.Wo * 9 * c8n;
  const bf16x8 zero8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % c8n);
    const int64_t r = idx / c8n;
    const int tap = (int)(r % 9);
    const int64_t pix = r / 9;
    const int64_t wo = pix % g.Wo, t = pix / g.Wo, ho = t % g.Ho, b = t / g.Ho;
    const int64_t hi = ho * g.sh + tap / 3 - g.pt, wi = wo * g.sw + tap % 3 - g.pl;
    const bool ok = hi >= 0 && hi < g.H && wi >= 0 && wi < g.W;
    *reinterpret_cast<bf16x8*>(col + idx * 8) = ok ? *reinterpret_cast<const bf16x8*>(x + ((b * g.H + hi) * g.W + wi) * g.Cin + c8 * 8) : zero8;
  }
}

template <int NJ, int NC, bool DX>
static bool launch_mfma(const bf16_t* src, const float* w, const float* bias, const float* gamma, const float* beta, bf16_t* d0, bf16_t* d1,
                        float* mean, float* rstd, const ConvGeom& g, hipStream_t s) {
  constexpr size_t lds = mfma_lds_bytes<NJ, NC, DX>();
  static_assert(lds <= 160 * 1024, "LDS image of the weights does not fit");
  const int64_t M = DX ? g.B * g.H * g.W : g.B * g.Ho * g.Wo;
  const int64_t ntiles = cdiv64(M, 16 * NWV);
  const int grid = (int)(ntiles < 256 ? ntiles : 256);
  // more than the 64 KiB default of dynamic LDS needs the attribute; refused -> the caller runs the general kernel
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_mfma_kernel<NJ, NC, DX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  conv3x3_mfma_kernel<NJ, NC, DX><<<grid, NTH, lds, s>>>(src, w, bias, gamma, beta, d0, d1, mean, rstd, g);
  return true;
}
template <int NJ>
static bool launch_gather(const bf16_t* x, const float* w, const float* bias, const float* gamma, const float* beta, bf16_t* d0, bf16_t* d1,
                          float* mean, float* rstd, const ConvGeom& g, hipStream_t s) {
  constexpr size_t lds = mfma_lds_bytes<NJ, 1, false, true>();
  const int64_t ntiles = cdiv64(g.B * g.Ho * g.Wo, 16 * NWV);
  const int grid = (int)(ntiles < 1024 ? ntiles : 1024);       // 4 workgroups per CU: the kernel is all load latency and stores
  static_assert(lds <= 160 * 1024, "LDS image of the weights does not fit");
  // NJ = 8 (Cout = 128) needs ~79 KiB, above the 64 KiB default of dynamic LDS
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_mfma_kernel<NJ, 1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  conv3x3_mfma_kernel<NJ, 1, false, true><<<grid, NTH, lds, s>>>(x, w, bias, gamma, beta, d0, d1, mean, rstd, g);
  return true;
}
template <bool DX>
static bool dispatch_mfma(int n_ch, int red_ch, const bf16_t* src, const float* w, const float* bias, const float* gamma, const float* beta,
                          bf16_t* d0, bf16_t* d1, float* mean, float* rstd, const ConvGeom& g, hipStream_t s) {
#define CV_CASE(n, c) if (n_ch == 16 * n && red_ch == 32 * c) return launch_mfma<n, c, DX>(src, w, bias, gamma, beta, d0, d1, mean, rstd, g, s);
  CV_CASE(4, 2) CV_CASE(2, 1) CV_CASE(2, 2) CV_CASE(4, 1)
#undef CV_CASE
  return false;
}
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline bool fits32(const ConvGeom& g) { return g.H < (1 << 24) && g.W < (1 << 24); }

}  // namespace cv

bool neurst_conv_fwd_fast(const bf16_t* x, const float* w, const float* bias, const float* gamma, const float* beta, bf16_t* conv_out,
                          bf16_t* out, float* mean, float* rstd, const ConvGeom& g, hipStream_t s) {
  if (!cv::fits32(g)) return false;
  if (g.kh == 3 && g.kw == 3 && cv::aligned16(x) && cv::aligned16(conv_out) && cv::aligned16(out) &&
      cv::dispatch_mfma<false>((int)g.Cout, (int)g.Cin, x, w, bias, gamma, beta, conv_out, out, mean, rstd, g, s))
    return true;
  const int K = g.kh * g.kw * (int)g.Cin;
  if (K <= 32 && cv::aligned16(conv_out) && cv::aligned16(out) && (int64_t)g.kh * g.W * g.Cin < (1 << 30)) {
    if (g.Cout == 64) return cv::launch_gather<4>(x, w, bias, gamma, beta, conv_out, out, mean, rstd, g, s);
    if (g.Cout == 32) return cv::launch_gather<2>(x, w, bias, gamma, beta, conv_out, out, mean, rstd, g, s);
    if (g.Cout == 128) return cv::launch_gather<8>(x, w, bias, gamma, beta, conv_out, out, mean, rstd, g, s);
  }
  if (K <= 32 && g.Cout <= 64) {
    const int64_t npix = g.B * g.Ho * g.Wo;
    const int64_t want = cdiv64(npix, 16);
    const int grid = (int)(want > 2048 ? 2048 : (want < 1 ? 1 : want));
    if (K <= 9) cv::conv_smallk_fwd_kernel<9><<<grid, 256, 0, s>>>(x, w, bias, gamma, beta, conv_out, out, mean, rstd, g, K);
    else if (K <= 27) cv::conv_smallk_fwd_kernel<27><<<grid, 256, 0, s>>>(x, w, bias, gamma, beta, conv_out, out, mean, rstd, g, K);
    else cv::conv_smallk_fwd_kernel<32><<<grid, 256, 0, s>>>(x, w, bias, gamma, beta, conv_out, out, mean, rstd, g, K);
    return true;
  }
  return false;
}

bool neurst_conv_bwd_pix_fast(const bf16_t* dout, const bf16_t* conv_out, const bf16_t* out, const float* gamma, const float* mean,
                              const float* rstd, bf16_t* dconv, float* dgamma, float* dbeta, float* dbias, const ConvGeom& g, hipStream_t s) {
  if (!(cv::aligned16(dout) && cv::aligned16(conv_out) && cv::aligned16(out) && cv::aligned16(dconv))) return false;
  const int64_t npix = g.B * g.Ho * g.Wo;
  const int lpp = (int)g.Cout / 8;
  if (g.Cout % 8 || (lpp != 2 && lpp != 4 && lpp != 8 && lpp != 16)) return false;
  const int64_t want = cdiv64(npix, 4 * (64 / lpp) * 4);
  const int grid = (int)(want > 512 ? 512 : (want < 1 ? 1 : want));
#define CV_PIX(l) cv::conv_bwd_pix_vec_kernel<l><<<grid, 256, 0, s>>>(dout, conv_out, out, gamma, mean, rstd, dconv, dgamma, dbeta, dbias, npix)
  if (lpp == 2) CV_PIX(2); else if (lpp == 4) CV_PIX(4); else if (lpp == 8) CV_PIX(8); else CV_PIX(16);
#undef CV_PIX
  return true;
}

bool neurst_conv_dx_fast(const bf16_t* dconv, const float* w, bf16_t* dx, const ConvGeom& g, hipStream_t s) {
  if (!cv::fits32(g) || g.kh != 3 || g.kw != 3 || !cv::aligned16(dconv) || !cv::aligned16(dx)) return false;
  return cv::dispatch_mfma<true>((int)g.Cin, (int)g.Cout, dconv, w, nullptr, nullptr, nullptr, dx, nullptr, nullptr, nullptr, g, s);
}

int64_t neurst_conv_col_bytes(const ConvGeom& g) {
  if (g.kh != 3 || g.kw != 3 || g.Cin % 8 || g.Cout % 8) return 0;
  return g.B * g.Ho * g.Wo * 9 * g.Cin * 2;
}

bool neurst_conv_dw_fast(const bf16_t* x, const bf16_t* dconv, float* dw, void* col_ws, int64_t col_ws_bytes, const ConvGeom& g, hipStream_t s) {
  const int K = g.kh * g.kw * (int)g.Cin;
  const int64_t npix = g.B * g.Ho * g.Wo;
  if (K <= 32 && g.Cout <= 64) {
    const int64_t want = cdiv64(npix, 16 * 8 * 4);
    const int grid = (int)(want > 512 ? 512 : (want < 1 ? 1 : want));
#define CV_DW(km)                                                                                                          \
  do {                                                                                                                       \
    constexpr int lds = 16 * km * 64 * 4;                                                                                    \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(cv::conv_smallk_dw_kernel<km>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { \
      (void)hipGetLastError();                                                                                               \
      return false;                                                                                                          \
    }                                                                                                                        \
    cv::conv_smallk_dw_kernel<km><<<grid, 1024, lds, s>>>(x, dconv, dw, g, K);                                               \
  } while (0)
    if (K <= 9) CV_DW(9); else if (K <= 27) CV_DW(27); else CV_DW(32);
#undef CV_DW
    return true;
  }
  const int64_t need = neurst_conv_col_bytes(g);
  if (need == 0 || !col_ws || col_ws_bytes < need || !cv::aligned16(x) || !cv::aligned16(col_ws) || !cv::aligned16(dconv)) return false;
  const int64_t chunks = need / 16;
  const int grid = (int)(cdiv64(chunks, 256) > 16384 ? 16384 : cdiv64(chunks, 256));
  cv::im2col3x3_kernel<<<grid, 256, 0, s>>>(x, reinterpret_cast<bf16_t*>(col_ws), g);
  // dw[K][Cout] += col^T . dconv, reduction over the pixels
  int64_t split = npix / 4096;
  split = split < 1 ? 1 : (split > 64 ? 64 : split);
  return neurst_gemm_bias_act(col_ws, dconv, dw, nullptr, nullptr, K, g.Cout, npix, K, g.Cout, g.Cout, 1, 0, NEURST_BF16, NEURST_F32, 0, 1,
                              (int)split, (void*)s) == NEURST_OK;
}
