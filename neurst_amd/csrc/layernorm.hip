// a7 (+a6): LN(x + dropout(y)) forward / backward.  HBM-bound: one wave per row, the
// row lives in registers (16 B per lane per access), statistics in fp32.
//   forward  bytes/row: read x, y; write z (over y), out          = 4*C*elt
//   backward bytes/row: read dout, z; write dz (+dy if dropout)    = 3..4*C*elt
// Reference: st/models/tools/utils.py:324-337, st/layers/common_layers_v1.py:342-347.
#include "common.h"

#define LN_EPS 1e-6f

template <typename T, int VEC> struct VecIO;
template <> struct VecIO<float, 4> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
};
template <> struct VecIO<bf16_t, 8> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (bf16_t)v[j];
    *reinterpret_cast<bf16x8*>(p) = t;
  }
};

// ------------------------------------------------------------------ forward (vector path)
template <typename T, int VEC, int NCH>
__global__ void __launch_bounds__(256) ln_fwd_vec(const T* __restrict__ x, T* __restrict__ yz, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, T* __restrict__ out, float* __restrict__ mean,
                                                   float* __restrict__ rstd, int64_t rows, int C, uint32_t thr, float scale,
                                                   uint32_t key, const int32_t* __restrict__ lens, int64_t Tn) {
  const int lane = threadIdx.x & 63;
  const int nchunks = C / VEC;
  const float invC = 1.f / (float)C;
  float gv[NCH][VEC], bv[NCH][VEC];   // row invariant: fetched once per wave
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + 64 * c;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      gv[c][j] = ch < nchunks ? gamma[ch * VEC + j] : 0.f;
      bv[c][j] = ch < nchunks ? beta[ch * VEC + j] : 0.f;
    }
  }
  for (int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    float z[NCH][VEC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int ch = lane + 64 * c;
      if (ch < nchunks) {
        float xv[VEC], yv[VEC];
        int64_t off = row * C + (int64_t)ch * VEC;
        VecIO<T, VEC>::load(x + off, xv);
        VecIO<T, VEC>::load(yz + off, yv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float yy = yv[j];
          if (thr) yy = dropout_keep(key, (uint64_t)(off + j), thr) ? yy * scale : 0.f;
          // the residual sum is rounded to the activation type, as it would be if stored
          z[c][j] = to_f32<T>(from_f32<T>(xv[j] + yy));
          s += z[c][j];
        }
        VecIO<T, VEC>::store(yz + off, z[c]);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) z[c][j] = 0.f;
      }
    }
    const float mu = wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (lane + 64 * c < nchunks) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { float d = z[c][j] - mu; q += d * d; }
      }
    const float var = wave_sum(q) * invC;
    const float rs = 1.0f / sqrtf(var + LN_EPS);
    bool keep_row = true;
    if (lens) keep_row = (row % Tn) < lens[row / Tn];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int ch = lane + 64 * c;
      if (ch < nchunks) {
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          o[j] = keep_row ? ((z[c][j] - mu) * rs) * gv[c][j] + bv[c][j] : 0.f;
        VecIO<T, VEC>::store(out + row * C + (int64_t)ch * VEC, o);
      }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// ------------------------------------------------------------------ forward (any C, scalar)
template <typename T>
__global__ void ln_fwd_generic(const T* __restrict__ x, T* __restrict__ yz, const float* __restrict__ gamma,
                               const float* __restrict__ beta, T* __restrict__ out, float* __restrict__ mean,
                               float* __restrict__ rstd, int64_t rows, int C, uint32_t thr, float scale, uint32_t key,
                               const int32_t* __restrict__ lens, int64_t Tn) {
  const int lane = threadIdx.x & 63;
  const float invC = 1.f / (float)C;
  for (int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
      int64_t off = row * C + c;
      float yy = to_f32<T>(yz[off]);
      if (thr) yy = dropout_keep(key, (uint64_t)off, thr) ? yy * scale : 0.f;
      T zt = from_f32<T>(to_f32<T>(x[off]) + yy);
      yz[off] = zt;
      s += to_f32<T>(zt);
    }
    const float mu = wave_sum(s) * invC;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { float d = to_f32<T>(yz[row * C + c]) - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum(q) * invC + LN_EPS);
    bool keep_row = true;
    if (lens) keep_row = (row % Tn) < lens[row / Tn];
    for (int c = lane; c < C; c += 64) {
      float zz = to_f32<T>(yz[row * C + c]);
      out[row * C + c] = from_f32<T>(keep_row ? ((zz - mu) * rs) * gamma[c] + beta[c] : 0.f);
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// ------------------------------------------------------------------ backward (vector path)
template <typename T, int VEC, int NCH>
__global__ void __launch_bounds__(256) ln_bwd_vec(const T* __restrict__ dout, const T* __restrict__ z, const float* __restrict__ gamma,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dz,
                                                   T* __restrict__ dy, float* __restrict__ partial, int64_t rows, int C, uint32_t thr,
                                                   float scale, uint32_t key, const int32_t* __restrict__ lens, int64_t Tn) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sm = reinterpret_cast<float*>(smem_raw);  // [2][4][C]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nchunks = C / VEC;
  const float invC = 1.f / (float)C;
  float dgam[NCH][VEC], dbet[NCH][VEC], gv[NCH][VEC];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    int ch = lane + 64 * c;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      dgam[c][j] = 0.f; dbet[c][j] = 0.f;
      gv[c][j] = ch < nchunks ? gamma[ch * VEC + j] : 0.f;
    }
  }
  for (int64_t row = blockIdx.x * 4 + w; row < rows; row += (int64_t)gridDim.x * 4) {
    bool keep_row = true;
    if (lens) keep_row = (row % Tn) < lens[row / Tn];
    const float mu = mean[row], rs = rstd[row];
    float g[NCH][VEC], xh[NCH][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int ch = lane + 64 * c;
      if (ch < nchunks) {
        float dv[VEC], zv[VEC];
        int64_t off = row * C + (int64_t)ch * VEC;
        VecIO<T, VEC>::load(dout + off, dv);
        VecIO<T, VEC>::load(z + off, zv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float d = keep_row ? dv[j] : 0.f;
          xh[c][j] = (zv[j] - mu) * rs;
          dgam[c][j] += d * xh[c][j];
          dbet[c][j] += d;
          g[c][j] = d * gv[c][j];
          s1 += g[c][j];
          s2 += g[c][j] * xh[c][j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { g[c][j] = 0.f; xh[c][j] = 0.f; }
      }
    }
    const float m1 = wave_sum(s1) * invC, m2 = wave_sum(s2) * invC;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int ch = lane + 64 * c;
      if (ch < nchunks) {
        float o[VEC];
        int64_t off = row * C + (int64_t)ch * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = rs * (g[c][j] - m1 - xh[c][j] * m2);
        VecIO<T, VEC>::store(dz + off, o);
        if (thr) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) o[j] = dropout_keep(key, (uint64_t)(off + j), thr) ? o[j] * scale : 0.f;
          VecIO<T, VEC>::store(dy + off, o);
        }
      }
    }
  }
  // block reduction of the per-wave column partials -> partial[blk][2][C]
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    int ch = lane + 64 * c;
    if (ch < nchunks) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        sm[(0 * 4 + w) * C + ch * VEC + j] = dgam[c][j];
        sm[(1 * 4 + w) * C + ch * VEC + j] = dbet[c][j];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    int which = i / C, col = i % C;
    float s = sm[(which * 4 + 0) * C + col] + sm[(which * 4 + 1) * C + col] + sm[(which * 4 + 2) * C + col] +
              sm[(which * 4 + 3) * C + col];
    partial[((int64_t)blockIdx.x * 2 + which) * C + col] = s;
  }
}

// out[which][col] += sum_blk partial[blk][which][col]; one block = 64 columns x 16 row lanes
__global__ void __launch_bounds__(1024) ln_bwd_reduce(const float* __restrict__ partial, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      int nblk, int C) {
  __shared__ float sm[16][64];
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;          // index into [2][C]
  float s = 0.f;
  if (i < 2 * C) {
    const int which = i / C, col = i % C;
    for (int b = rl; b < nblk; b += 16) s += partial[((int64_t)b * 2 + which) * C + col];
  }
  sm[rl][lane] = s;
  __syncthreads();
  if (rl == 0 && i < 2 * C) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += sm[r][lane];
    if (i < C) dgamma[i] += t; else dbeta[i - C] += t;
  }
}

template <typename T>
__global__ void ln_bwd_generic(const T* __restrict__ dout, const T* __restrict__ z, const float* __restrict__ gamma,
                               const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dz, T* __restrict__ dy,
                               float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int C, uint32_t thr, float scale,
                               uint32_t key, const int32_t* __restrict__ lens, int64_t Tn) {
  const int lane = threadIdx.x & 63;
  const float invC = 1.f / (float)C;
  for (int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    bool keep_row = true;
    if (lens) keep_row = (row % Tn) < lens[row / Tn];
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
      float d = keep_row ? to_f32<T>(dout[row * C + c]) : 0.f;
      float xh = (to_f32<T>(z[row * C + c]) - mu) * rs;
      float g = d * gamma[c];
      s1 += g; s2 += g * xh;
      atomicAdd(dgamma + c, d * xh);
      atomicAdd(dbeta + c, d);
    }
    const float m1 = wave_sum(s1) * invC, m2 = wave_sum(s2) * invC;
    for (int c = lane; c < C; c += 64) {
      int64_t off = row * C + c;
      float d = keep_row ? to_f32<T>(dout[off]) : 0.f;
      float xh = (to_f32<T>(z[off]) - mu) * rs;
      float o = rs * (d * gamma[c] - m1 - xh * m2);
      dz[off] = from_f32<T>(o);
      if (thr) dy[off] = from_f32<T>(dropout_keep(key, (uint64_t)off, thr) ? o * scale : 0.f);
    }
  }
}

// ------------------------------------------------------------------ host side
static inline int ln_nch(int64_t C, int vec) { return (int)cdiv64(C / vec, 64); }
static inline bool ln_vec_ok(int64_t C, int vec, const void* a, const void* b, const void* c) {
  return C % vec == 0 && ln_nch(C, vec) <= 8 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0) && C % 4 == 0;
}
extern "C" int64_t neurst_layernorm_bwd_blocks(int64_t rows) {
  int64_t g = cdiv64(rows, 4);
  return g > 512 ? 512 : (g < 1 ? 1 : g);
}

extern "C" int neurst_dropout_add_layernorm_fwd(const void* x, void* yz, const float* gamma, const float* beta, void* out,
                                                float* mean, float* rstd, int64_t rows, int64_t C, float rate, uint32_t key,
                                                const int32_t* lens, int64_t T, int dtype, void* stream) {
  NEURST_CHECK_ARG(rows > 0 && C > 0 && rate >= 0.f && rate < 1.f, "layernorm_fwd: bad args");
  NEURST_CHECK_ARG(!lens || T > 0, "layernorm_fwd: lens given without T");
  uint32_t thr = rate > 0.f ? rate_threshold(rate) : 0u;
  float sc = keep_scale(rate);
  int64_t g = cdiv64(rows, 4);
  int grid = (int)(g > 4096 ? 4096 : g);
  hipStream_t s = as_stream(stream);
#define FWD_VEC(TT, VEC, NCH)                                                                                            \
  ln_fwd_vec<TT, VEC, NCH><<<grid, 256, 0, s>>>((const TT*)x, (TT*)yz, gamma, beta, (TT*)out, mean, rstd, rows, (int)C, \
                                                thr, sc, key, lens, T)
#define FWD_DISPATCH(TT, VEC)                                         \
  switch (ln_nch(C, VEC)) {                                           \
    case 1: FWD_VEC(TT, VEC, 1); break;                               \
    case 2: FWD_VEC(TT, VEC, 2); break;                               \
    case 3: FWD_VEC(TT, VEC, 3); break;                               \
    case 4: FWD_VEC(TT, VEC, 4); break;                               \
    default: FWD_VEC(TT, VEC, 8); break;                              \
  }
  if (dtype == NEURST_BF16) {
    if (ln_vec_ok(C, 8, x, yz, out)) { FWD_DISPATCH(bf16_t, 8) }
    else ln_fwd_generic<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)yz, gamma, beta, (bf16_t*)out, mean, rstd, rows, (int)C, thr, sc, key, lens, T);
  } else {
    if (ln_vec_ok(C, 4, x, yz, out)) { FWD_DISPATCH(float, 4) }
    else ln_fwd_generic<float><<<grid, 256, 0, s>>>((const float*)x, (float*)yz, gamma, beta, (float*)out, mean, rstd, rows, (int)C, thr, sc, key, lens, T);
  }
#undef FWD_DISPATCH
#undef FWD_VEC
  NEURST_CHECK_LAUNCH("dropout_add_layernorm_fwd");
  return NEURST_OK;
}

extern "C" int neurst_dropout_add_layernorm_bwd(const void* dout, const void* z, const float* gamma, const float* mean,
                                                const float* rstd, void* dz, void* dy, float* dgamma, float* dbeta, float* partial,
                                                int64_t rows, int64_t C, float rate, uint32_t key, const int32_t* lens, int64_t T,
                                                int dtype, void* stream) {
  NEURST_CHECK_ARG(rows > 0 && C > 0 && rate >= 0.f && rate < 1.f, "layernorm_bwd: bad args");
  NEURST_CHECK_ARG(rate == 0.f || (dy && dy != dz), "layernorm_bwd: dropout needs a separate dy buffer");
  NEURST_CHECK_ARG(!lens || T > 0, "layernorm_bwd: lens given without T");
  uint32_t thr = rate > 0.f ? rate_threshold(rate) : 0u;
  float sc = keep_scale(rate);
  int grid = (int)neurst_layernorm_bwd_blocks(rows);
  hipStream_t s = as_stream(stream);
  size_t shmem = (size_t)8 * C * sizeof(float);
  bool vec = false;
#define BWD_VEC(TT, VEC, NCH)                                                                                                \
  ln_bwd_vec<TT, VEC, NCH><<<grid, 256, shmem, s>>>((const TT*)dout, (const TT*)z, gamma, mean, rstd, (TT*)dz, (TT*)dy, partial, \
                                                    rows, (int)C, thr, sc, key, lens, T)
#define BWD_DISPATCH(TT, VEC)                                         \
  switch (ln_nch(C, VEC)) {                                           \
    case 1: BWD_VEC(TT, VEC, 1); break;                               \
    case 2: BWD_VEC(TT, VEC, 2); break;                               \
    case 3: BWD_VEC(TT, VEC, 3); break;                               \
    case 4: BWD_VEC(TT, VEC, 4); break;                               \
    default: BWD_VEC(TT, VEC, 8); break;                              \
  }
  if (dtype == NEURST_BF16) {
    if (ln_vec_ok(C, 8, dout, z, dz) && (!thr || ((uintptr_t)dy & 15) == 0) && partial && shmem <= 64 * 1024) { vec = true; BWD_DISPATCH(bf16_t, 8) }
    else ln_bwd_generic<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)dout, (const bf16_t*)z, gamma, mean, rstd, (bf16_t*)dz, (bf16_t*)dy, dgamma, dbeta, rows, (int)C, thr, sc, key, lens, T);
  } else {
    if (ln_vec_ok(C, 4, dout, z, dz) && (!thr || ((uintptr_t)dy & 15) == 0) && partial && shmem <= 64 * 1024) { vec = true; BWD_DISPATCH(float, 4) }
    else ln_bwd_generic<float><<<grid, 256, 0, s>>>((const float*)dout, (const float*)z, gamma, mean, rstd, (float*)dz, (float*)dy, dgamma, dbeta, rows, (int)C, thr, sc, key, lens, T);
  }
#undef BWD_DISPATCH
#undef BWD_VEC
  NEURST_CHECK_LAUNCH("dropout_add_layernorm_bwd");
  if (vec) {
    ln_bwd_reduce<<<(int)cdiv64(2 * C, 64), 1024, 0, s>>>(partial, dgamma, dbeta, grid, (int)C);
    NEURST_CHECK_LAUNCH("layernorm_bwd_reduce");
  }
  return NEURST_OK;
}
