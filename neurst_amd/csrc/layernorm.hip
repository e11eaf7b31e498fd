// a7 (+a6): LN(x + dropout(y)) forward / backward.  HBM-bound: one wave per row, the
// row lives in registers (16 B per lane per access), statistics in fp32.
//   forward  bytes/row: read x, y; write z (over y), out          = 4*C*elt
//   backward bytes/row: read dout, z; write dz (+dy if dropout)    = 3..4*C*elt
// Reference: st/models/tools/utils.py:324-337, st/layers/common_layers_v1.py:342-347.
#include "common.h"

#define LN_EPS 1e-6f

template <typename T, int VEC> struct VecIO;
template <> struct VecIO<float, 4> {
  typedef f32x4 Raw;
  static __device__ __forceinline__ Raw load_raw(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void unpack(const Raw& t, float (&v)[4]) { v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
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
  typedef bf16x8 Raw;
  static __device__ __forceinline__ Raw load_raw(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
  static __device__ __forceinline__ void unpack(const Raw& t, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
  }
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

// 8-byte lanes: a 768-wide bf16 row is 192 chunks of 4 = three full wave passes, where 16-byte lanes give 96 chunks = one and a
// half passes (half the wave idles in the second) and cost a third more registers per lane for the same row
template <> struct VecIO<bf16_t, 4> {
  typedef bf16x4 Raw;
  static __device__ __forceinline__ Raw load_raw(const bf16_t* p) { return *reinterpret_cast<const bf16x4*>(p); }
  static __device__ __forceinline__ void unpack(const Raw& t, float (&v)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)t[j];
  }
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)t[j];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
    bf16x4 t;
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = (bf16_t)v[j];
    *reinterpret_cast<bf16x4*>(p) = t;
  }
};

// row-invariant parameter chunk (VEC consecutive floats) with 16-byte loads when the pointer allows
template <int VEC>
__device__ __forceinline__ void load_param(const float* __restrict__ p, int ch, bool valid, bool vec_ok, float (&v)[VEC]) {
  if (valid && vec_ok) {
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p + ch * VEC + 4 * q);
      v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
    }
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = valid ? p[ch * VEC + j] : 0.f;
  }
}

// dropout decision for flat element index idx; idx32: every index of the tensor fits 32 bits, so the inner hash
// fmix32(hi ^ key) is the launch constant `inner0`
__device__ __forceinline__ bool ln_keep(uint32_t key, uint32_t inner0, bool idx32, int64_t idx, uint32_t thr) {
  if (idx32) return (fmix32((uint32_t)idx ^ inner0) >> 8) >= thr;
  return dropout_keep(key, (uint64_t)idx, thr);
}

// ------------------------------------------------------------------ forward (vector path)
// One wave owns R rows at a time (all their loads are issued before the first reduction, so R rows of
// bytes are in flight per wave); a row lives in registers as NCH 16-byte chunks per lane.
template <typename T, int VEC, int NCH, int R, bool PF>
__global__ void __launch_bounds__(256) ln_fwd_vec(const T* __restrict__ x, T* __restrict__ yz, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, T* __restrict__ out, float* __restrict__ mean,
                                                   float* __restrict__ rstd, int64_t rows, int C, uint32_t thr, float scale,
                                                   uint32_t key, uint32_t inner0, int idx32, const int32_t* __restrict__ lens, int64_t Tn) {
  const int lane = threadIdx.x & 63;
  const int nchunks = C / VEC;
  const float invC = 1.f / (float)C;
  float gv[NCH][VEC], bv[NCH][VEC];   // row invariant: fetched once per wave
  const bool pvec = ((((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + 64 * c;
    load_param<VEC>(gamma, ch, ch < nchunks, pvec, gv[c]);
    load_param<VEC>(beta, ch, ch < nchunks, pvec, bv[c]);
  }
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  // PF (one row per wave): the NEXT row's x and y are requested, as loaded (packed), before this row's stores go out.  The
  // vector-memory counter is one in-order count of loads and stores: loads issued behind a row's stores cannot be waited
  // for without waiting for those stores too, so every row used to cost a load AND a store round trip, one after the other.
  typename VecIO<T, VEC>::Raw px[NCH], py[NCH];
  if constexpr (PF) {
    static_assert(R == 1, "the prefetching loop handles one row per wave");
    const int64_t row = wave_id;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunks && row < rows) {
        px[c] = VecIO<T, VEC>::load_raw(x + row * C + (int64_t)ch * VEC);
        py[c] = VecIO<T, VEC>::load_raw(yz + row * C + (int64_t)ch * VEC);
      }
    }
  }
  for (int64_t row0 = wave_id * R; row0 < rows; row0 += nwaves * R) {
    float xv[R][NCH][VEC], z[R][NCH][VEC];
    if constexpr (PF) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (lane + 64 * c < nchunks) {
          VecIO<T, VEC>::unpack(px[c], xv[0][c]);
          VecIO<T, VEC>::unpack(py[c], z[0][c]);
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) { xv[0][c][j] = 0.f; z[0][c][j] = 0.f; }
        }
      }
      const int64_t nrow = row0 + nwaves;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunks && nrow < rows) {
          px[c] = VecIO<T, VEC>::load_raw(x + nrow * C + (int64_t)ch * VEC);
          py[c] = VecIO<T, VEC>::load_raw(yz + nrow * C + (int64_t)ch * VEC);
        }
      }
    } else {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        const int64_t off = (row0 + r) * C + (int64_t)ch * VEC;
        if (ch < nchunks && row0 + r < rows) {
          VecIO<T, VEC>::load(x + off, xv[r][c]);
          VecIO<T, VEC>::load(yz + off, z[r][c]);
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) { xv[r][c][j] = 0.f; z[r][c][j] = 0.f; }
        }
      }
    }
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s[r] = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        const int64_t off = (row0 + r) * C + (int64_t)ch * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float yy = z[r][c][j];
          if (thr) yy = ln_keep(key, inner0, idx32, off + j, thr) ? yy * scale : 0.f;
          // the residual sum is rounded to the activation type, as it would be if stored
          z[r][c][j] = to_f32<T>(from_f32<T>(xv[r][c][j] + yy));
          s[r] += z[r][c][j];
        }
        if (ch < nchunks && row0 + r < rows) VecIO<T, VEC>::store(yz + off, z[r][c]);
      }
    }
    float mu[R], rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) mu[r] = wave_sum_dpp(s[r]) * invC;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (lane + 64 * c < nchunks) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) { float d = z[r][c][j] - mu[r]; q += d * d; }
        }
      s[r] = q;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) rs[r] = 1.0f / sqrtf(wave_sum_dpp(s[r]) * invC + LN_EPS);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = row0 + r;
      if (row >= rows) break;
      bool keep_row = true;
      if (lens) keep_row = (row % Tn) < lens[row / Tn];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunks) {
          float o[VEC];
#pragma unroll
          for (int j = 0; j < VEC; ++j)
            o[j] = keep_row ? ((z[r][c][j] - mu[r]) * rs[r]) * gv[c][j] + bv[c][j] : 0.f;
          VecIO<T, VEC>::store(out + row * C + (int64_t)ch * VEC, o);
        }
      }
      if (lane == 0) { mean[row] = mu[r]; rstd[row] = rs[r]; }
    }
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
// DB: also dbias[col] += sum over rows of dy (the gradient of the bias of the dense layer that produced y: the weight-
// gradient GEMM of that layer then needs no operand sum)
template <typename T, int VEC, int NCH, bool DB, bool PF>
__global__ void __launch_bounds__(1024) ln_bwd_vec(const T* __restrict__ dout, const T* __restrict__ z, const float* __restrict__ gamma,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dz,
                                                   T* __restrict__ dy, float* __restrict__ partial, float* __restrict__ dgamma,
                                                   float* __restrict__ dbeta, float* __restrict__ dbias, int64_t rows, int C, uint32_t thr,
                                                   float scale, uint32_t key, uint32_t inner0, int idx32,
                                                   const int32_t* __restrict__ lens, int64_t Tn) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sm = reinterpret_cast<float*>(smem_raw);  // [2 or 3][nw][C]
  const int nw = blockDim.x >> 6;           // 4 or 8 waves: one row per wave in flight, so more waves = more rows in flight
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nchunks = C / VEC;
  const float invC = 1.f / (float)C;
  float dgam[NCH][VEC], dbet[NCH][VEC], gv[NCH][VEC], dbia[DB ? NCH : 1][VEC];
  const bool pvec = (((uintptr_t)gamma) & 15) == 0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    int ch = lane + 64 * c;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { dgam[c][j] = 0.f; dbet[c][j] = 0.f; if constexpr (DB) dbia[c][j] = 0.f; }
    load_param<VEC>(gamma, ch, ch < nchunks, pvec, gv[c]);
  }
  // PF: the next row's dout and z are requested (packed, as loaded) before this row's stores go out - see ln_fwd_vec
  typename VecIO<T, VEC>::Raw pd[NCH], pz[NCH];
  const int64_t rstride = (int64_t)gridDim.x * nw;
  if constexpr (PF) {
    const int64_t row = (int64_t)blockIdx.x * nw + w;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunks && row < rows) {
        pd[c] = VecIO<T, VEC>::load_raw(dout + row * C + (int64_t)ch * VEC);
        pz[c] = VecIO<T, VEC>::load_raw(z + row * C + (int64_t)ch * VEC);
      }
    }
  }
  for (int64_t row = (int64_t)blockIdx.x * nw + w; row < rows; row += rstride) {
    bool keep_row = true;
    if (lens) keep_row = (row % Tn) < lens[row / Tn];
    const float mu = mean[row], rs = rstd[row];
    // PF keeps the row PACKED (as loaded) and converts it a second time for the output pass instead of holding g and x^
    // in fp32: 2 x NCH x 4 registers instead of 2 x NCH x VEC (the 16-wave workgroup has 128 registers per lane)
    float g[PF ? 1 : NCH][VEC], xh[PF ? 1 : NCH][VEC];
    typename VecIO<T, VEC>::Raw cd[NCH], cz[NCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int ch = lane + 64 * c;
      if (ch < nchunks) {
        float dv[VEC], zv[VEC];
        int64_t off = row * C + (int64_t)ch * VEC;
        if constexpr (PF) {
          cd[c] = pd[c];
          cz[c] = pz[c];
          VecIO<T, VEC>::unpack(cd[c], dv);
          VecIO<T, VEC>::unpack(cz[c], zv);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float d = keep_row ? dv[j] : 0.f, x = (zv[j] - mu) * rs, gg = d * gv[c][j];
            dgam[c][j] += d * x;
            dbet[c][j] += d;
            s1 += gg;
            s2 += gg * x;
          }
        } else {
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
        }
      } else if constexpr (!PF) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { g[c][j] = 0.f; xh[c][j] = 0.f; }
      }
    }
    if constexpr (PF) {
      const int64_t nrow = row + rstride;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunks && nrow < rows) {
          pd[c] = VecIO<T, VEC>::load_raw(dout + nrow * C + (int64_t)ch * VEC);
          pz[c] = VecIO<T, VEC>::load_raw(z + nrow * C + (int64_t)ch * VEC);
        }
      }
    }
    const float m1 = wave_sum_dpp(s1) * invC, m2 = wave_sum_dpp(s2) * invC;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int ch = lane + 64 * c;
      if (ch < nchunks) {
        float o[VEC];
        int64_t off = row * C + (int64_t)ch * VEC;
        if constexpr (PF) {
          float dv[VEC], zv[VEC];
          VecIO<T, VEC>::unpack(cd[c], dv);
          VecIO<T, VEC>::unpack(cz[c], zv);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float d = keep_row ? dv[j] : 0.f, x = (zv[j] - mu) * rs, gg = d * gv[c][j];
            o[j] = rs * (gg - m1 - x * m2);
          }
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) o[j] = rs * (g[c][j] - m1 - xh[c][j] * m2);
        }
        VecIO<T, VEC>::store(dz + off, o);
        if (thr) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) o[j] = ln_keep(key, inner0, idx32, off + j, thr) ? o[j] * scale : 0.f;
          VecIO<T, VEC>::store(dy + off, o);
        }
        if constexpr (DB) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) dbia[c][j] += o[j];
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
        sm[(0 * nw + w) * C + ch * VEC + j] = dgam[c][j];
        sm[(1 * nw + w) * C + ch * VEC + j] = dbet[c][j];
        if constexpr (DB) sm[(2 * nw + w) * C + ch * VEC + j] = dbia[c][j];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (DB ? 3 : 2) * C; i += blockDim.x) {
    int which = i / C, col = i % C;
    float s = 0.f;
    for (int ww = 0; ww < nw; ++ww) s += sm[(which * nw + ww) * C + col];
    if (which == 2) atomicAdd(dbias + col, s);
    else if (partial) partial[((int64_t)blockIdx.x * 2 + which) * C + col] = s;
    else atomicAdd((which ? dbeta : dgamma) + col, s);   // a few hundred adds per address: cheaper than a second launch
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

// ------------------------------------------------------------------ straight-line kernels for the training shapes
// bf16 rows of exactly 64 NCH VEC columns (768 = 64 x 3 x 4, 512 = 64 x 1 x 8, 256 = 64 x 1 x 4), 32-bit element indices, no
// row mask, column sums by atomics: everything the general kernels decide per element or per chunk at run time (lane in range?
// dropout on? 32- or 64-bit hash? row masked?) is known here when the kernel is compiled, so a row is ONE basic block.  That
// matters beyond the branch instructions: hipcc places exact vmcnt waits only inside a basic block - across the general
// kernels' per-chunk branches every wait was vmcnt(0), i.e. the next row's loads also waited for this row's stores.
// The next row (and its mean / rstd) is requested before this row's stores go out; its index is clamped instead of tested.
// The next rows are requested by inline-assembly loads in program order and awaited by COUNTED waits written here (hipcc moved
// its own loads of the next row below the current row's work, or made their first use wait for everything): when a register
// set is first read, the only younger vector-memory operations are the other set's stores and its refill, and exactly those
// may stay outstanding.  The counts rely on the compiler issuing nothing else through the vector-memory counter inside the row
// loop - no scratch traffic (checked: private_segment size 0 for every instantiation) and one store per chunk.
typedef uint32_t __attribute__((ext_vector_type(2))) u32x2_t;
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4_t;
template <int VEC> struct LineRaw;
template <> struct LineRaw<4> {
  typedef bf16x4 T;
  typedef u32x2_t R;
  template <int OFF> static __device__ __forceinline__ void load(R& r, const bf16_t* p) {   // OFF: byte offset, an immediate
    asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
  }
};
template <> struct LineRaw<8> {
  typedef bf16x8 T;
  typedef u32x4_t R;
  template <int OFF> static __device__ __forceinline__ void load(R& r, const bf16_t* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
  }
};
// chunk c of a row sits 64 VEC elements behind chunk c - 1: one address register pair per tensor, the rest in the instruction
template <int VEC, int NCH, int c = 0, typename Raw>
__device__ __forceinline__ void line_fetch(Raw (&f)[NCH], const bf16_t* p) {
  if constexpr (c < NCH) {
    LineRaw<VEC>::template load<c * 64 * VEC * 2>(f[c], p);
    line_fetch<VEC, NCH, c + 1>(f, p);
  }
}
#define LN_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// (no __restrict__ on these kernels' pointers: with it, alias analysis lets the compiler move its stores across the inline-assembly
// loads - "they cannot touch the same memory" - and the counted waits would then count operations that are not there yet)
template <int VEC, int NCH, bool DROP>
__global__ void __launch_bounds__(256) ln_fwd_line(const bf16_t* x, bf16_t* yz, const float* gamma,
                                                    const float* beta, bf16_t* out, float* mean,
                                                    float* rstd, int64_t rows, uint32_t thr, float scale, uint32_t inner0) {
  typedef typename LineRaw<VEC>::T Vec;
  typedef typename LineRaw<VEC>::R Raw;
  constexpr int C = 64 * NCH * VEC;
  constexpr float invC = 1.f / (float)C;
  constexpr int NL = 2 * NCH, NS = 2 * NCH + 2;              // loads of a fetch; stores of a row (z, out, mean, rstd)
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float gv[NCH][VEC], bv[NCH][VEC];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      const f32x4 tg = *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * c) * VEC + 4 * q);
      const f32x4 tb = *reinterpret_cast<const f32x4*>(beta + (lane + 64 * c) * VEC + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gv[c][4 * q + e] = tg[e]; bv[c][4 * q + e] = tb[e]; }
    }
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int64_t row = (int64_t)blockIdx.x * 4 + w;
  if (row >= rows) return;
  Raw ax[NCH], ay[NCH], bx[NCH], by[NCH];                    // two register sets: one worked on, one being filled
  auto fetch = [&](Raw (&fx)[NCH], Raw (&fy)[NCH], int64_t r) __attribute__((always_inline)) {
    line_fetch<VEC, NCH>(fx, x + r * C + lane * VEC);
    line_fetch<VEC, NCH>(fy, yz + r * C + lane * VEC);
  };
  auto work = [&](Raw (&rx)[NCH], Raw (&ry)[NCH], int64_t r) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(rx[c]), "+v"(ry[c]));      // read only behind the wait above
    float z[NCH][VEC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const Vec cx = __builtin_bit_cast(Vec, rx[c]), cy = __builtin_bit_cast(Vec, ry[c]);
      const uint32_t off = (uint32_t)(r * C) + (lane + 64 * c) * VEC;
      Vec zr;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float yy = (float)cy[j];
        if constexpr (DROP) yy = (fmix32((off + j) ^ inner0) >> 8) >= thr ? yy * scale : 0.f;
        zr[j] = (bf16_t)((float)cx[j] + yy);                 // the residual sum is rounded to the activation type, as stored
        z[c][j] = (float)zr[j];
        s += z[c][j];
      }
      *reinterpret_cast<Vec*>(yz + r * C + (lane + 64 * c) * VEC) = zr;
    }
    const float mu = wave_sum_dpp(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < VEC; ++j) { const float d = z[c][j] - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum_dpp(q) * invC + LN_EPS);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      Vec o;
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = (bf16_t)(((z[c][j] - mu) * rs) * gv[c][j] + bv[c][j]);
      *reinterpret_cast<Vec*>(out + r * C + (lane + 64 * c) * VEC) = o;
    }
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
  };
  // rows of this wave: row + k nwaves, k < n; a request past the end asks for the last row again (harmless)
  const int64_t n = (rows - row + nwaves - 1) / nwaves;
  auto rowk = [&](int64_t k) { return row + (k < n ? k : n - 1) * nwaves; };
  fetch(ax, ay, row);
  fetch(bx, by, rowk(1));
  LN_WAIT(NL);                                               // the first row is here (gamma / beta were requested before it)
  work(ax, ay, row);
  int64_t k = 1;
  for (; k + 1 < n; k += 2) {
    fetch(ax, ay, rowk(k + 1));
    LN_WAIT(NS + NL);                                        // younger than set B: the stores of the row before, the refill of set A
    work(bx, by, rowk(k));
    fetch(bx, by, rowk(k + 2));
    LN_WAIT(NS + NL);
    work(ax, ay, rowk(k + 1));
  }
  if (k < n) {
    LN_WAIT(0);
    work(bx, by, rowk(k));
  }
  LN_WAIT(0);                                                // requests past the end still in flight: their registers are dead, but land them
}

template <int VEC, int NCH, bool DROP>
__global__ void __launch_bounds__(1024) ln_bwd_line(const bf16_t* dout, const bf16_t* z, const float* gamma,
                                                    const float* mean, const float* rstd, bf16_t* dz,
                                                    bf16_t* dy, float* dgamma, float* dbeta,
                                                    int64_t rows, uint32_t thr, float scale, uint32_t inner0) {
  typedef typename LineRaw<VEC>::T Vec;
  typedef typename LineRaw<VEC>::R Raw;
  constexpr int C = 64 * NCH * VEC;
  constexpr float invC = 1.f / (float)C;
  constexpr int NL = 2 * NCH + 2, NS = NCH * (DROP ? 2 : 1);   // loads of a fetch (dout, z, mean, rstd); stores of a row
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sm = reinterpret_cast<float*>(smem_raw);          // [2][nw][C]
  const int nw = blockDim.x >> 6;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float dgam[NCH][VEC], dbet[NCH][VEC], gv[NCH][VEC];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      const f32x4 tg = *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * c) * VEC + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gv[c][4 * q + e] = tg[e]; dgam[c][4 * q + e] = 0.f; dbet[c][4 * q + e] = 0.f; }
    }
  const int64_t rstride = (int64_t)gridDim.x * nw;
  const int64_t row = (int64_t)blockIdx.x * nw + w;
  if (row < rows) {
    Raw ad[NCH], az[NCH], bd[NCH], bz[NCH];
    uint32_t amu, ars, bmu, brs;                             // the row's mean and rstd ride with its fetch (a compiler-issued load
                                                             // here would be waited for with vmcnt(0): the refill with it)
    auto fetch = [&](Raw (&fd)[NCH], Raw (&fz)[NCH], uint32_t& fmu, uint32_t& frs, int64_t r) __attribute__((always_inline)) {
      asm volatile("global_load_dword %0, %1, off" : "=v"(fmu) : "v"(mean + r) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "=v"(frs) : "v"(rstd + r) : "memory");
      line_fetch<VEC, NCH>(fd, dout + r * C + lane * VEC);
      line_fetch<VEC, NCH>(fz, z + r * C + lane * VEC);
    };
    auto work = [&](Raw (&rd)[NCH], Raw (&rz)[NCH], uint32_t& rmu, uint32_t& rrs, int64_t r) __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(rd[c]), "+v"(rz[c]));
      asm volatile("" : "+v"(rmu), "+v"(rrs));
      const float mu = __builtin_bit_cast(float, rmu), rs = __builtin_bit_cast(float, rrs);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const Vec cd = __builtin_bit_cast(Vec, rd[c]), cz = __builtin_bit_cast(Vec, rz[c]);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float d = (float)cd[j], xh = ((float)cz[j] - mu) * rs, gg = d * gv[c][j];
          dgam[c][j] += d * xh;
          dbet[c][j] += d;
          s1 += gg;
          s2 += gg * xh;
        }
      }
      const float m1 = wave_sum_dpp(s1) * invC, m2 = wave_sum_dpp(s2) * invC;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const Vec cd = __builtin_bit_cast(Vec, rd[c]), cz = __builtin_bit_cast(Vec, rz[c]);
        const uint32_t off = (uint32_t)(r * C) + (lane + 64 * c) * VEC;
        Vec o, od;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float d = (float)cd[j], xh = ((float)cz[j] - mu) * rs, gg = d * gv[c][j];
          const float v = rs * (gg - m1 - xh * m2);
          o[j] = (bf16_t)v;
          if constexpr (DROP) od[j] = (bf16_t)((fmix32((off + j) ^ inner0) >> 8) >= thr ? v * scale : 0.f);
        }
        *reinterpret_cast<Vec*>(dz + r * C + (lane + 64 * c) * VEC) = o;
        if constexpr (DROP) *reinterpret_cast<Vec*>(dy + r * C + (lane + 64 * c) * VEC) = od;
      }
    };
    const int64_t n = (rows - row + rstride - 1) / rstride;  // see ln_fwd_line
    auto rowk = [&](int64_t k) { return row + (k < n ? k : n - 1) * rstride; };
    fetch(ad, az, amu, ars, row);
    fetch(bd, bz, bmu, brs, rowk(1));
    LN_WAIT(NL);
    work(ad, az, amu, ars, row);
    int64_t k = 1;
    for (; k + 1 < n; k += 2) {
      fetch(ad, az, amu, ars, rowk(k + 1));
      LN_WAIT(NS + NL);
      work(bd, bz, bmu, brs, rowk(k));
      fetch(bd, bz, bmu, brs, rowk(k + 2));
      LN_WAIT(NS + NL);
      work(ad, az, amu, ars, rowk(k + 1));
    }
    if (k < n) {
      LN_WAIT(0);
      work(bd, bz, bmu, brs, rowk(k));
    }
    LN_WAIT(0);
  }
  // block reduction of the per-wave column partials, then a few hundred atomic adds per address over the grid
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      sm[(0 * nw + w) * C + (lane + 64 * c) * VEC + j] = dgam[c][j];
      sm[(1 * nw + w) * C + (lane + 64 * c) * VEC + j] = dbet[c][j];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    const int which = i / C, col = i % C;
    float s = 0.f;
    for (int ww = 0; ww < nw; ++ww) s += sm[(which * nw + ww) * C + col];
    atomicAdd((which ? dbeta : dgamma) + col, s);
  }
}

// ------------------------------------------------------------------ host side
// neurst_set_option("ln_prefetch", bits): 1 = forward, 2 = backward row loop with the next row requested ahead of the stores.
// Measured on 16000 x 768 bf16, rate 0.1 (tools/bench_kernels.py ln, launch overhead included): forward 31.0 -> 29.6 us;
// backward 33.6 -> 37.6 us - its 16-wave workgroups get 128 registers per lane and the 16 registers of the packed next row
// push the column-sum accumulators into scratch inside the loop - so only the forward takes it by default.
int g_opt_ln_prefetch = [] { const char* e = getenv("NEURST_LN_PREFETCH"); return e ? atoi(e) : 53; }();
// bits 4 / 8 of the same option: forward / backward take 8-byte lanes when a bf16 row is whole wave passes of them but not of
// 16-byte lanes (C = 768: 3 x 64 x 4 against 1.5 x 64 x 8).  Measured at 16000 x 768, rate 0.1: forward 28.4 -> 26.8 us alone and
// 0.392 -> 0.373 ms per step; the backward 32.2 -> 30.9 us alone but 0.510 -> 0.537 ms inside the step (same box, alternating
// runs) - so the forward takes them by default (bits 1 + 4) and the backward does not.
static inline bool ln_lanes8(int64_t C, int bit) {
  return (g_opt_ln_prefetch & bit) && C % 256 == 0 && C % 512 != 0 && C / 256 <= 4;
}
static inline int ln_nch(int64_t C, int vec) { return (int)cdiv64(C / vec, 64); }
// bits 16 / 32: forward / backward of the training shapes through the straight-line kernels (ln_fwd_line / ln_bwd_line)
// (not 1024 = 64 x 2 x 8: its backward needs more than the 128 registers a 16-wave workgroup has, and scratch traffic would
// break the kernels' wait counts)
static inline bool ln_line_shape(int64_t C) { return C == 256 || C == 512 || C == 768; }
static inline bool ln_vec_ok(int64_t C, int vec, const void* a, const void* b, const void* c) {
  return C % vec == 0 && ln_nch(C, vec) <= 8 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0) && C % 4 == 0;
}
// Launch shape of ln_bwd_vec.  The tail of every workgroup adds 2 C column partials to dgamma / dbeta with atomics, so few,
// fat workgroups win: measured on 16000 x 768 (rate 0.1): 512 x 256 threads 42.8 us, 512 x 512 35.0, 256 x 1024 33.8
// (one workgroup of 16 waves per CU: 16 rows in flight per CU), 320 x 1024 42.6, 128 x 1024 44.1.
// Decoder-size inputs (1280 rows) as well: 80 workgroups of 16 waves instead of 320 of 4 - a quarter of the atomic adds on
// the same 2 C addresses; kernel alone 15 -> 6.6 us (NEURST_LN_BWD_SMALL_THREADS = 256 / 512 / 1024 for A/B timing).
static int ln_small_threads() {
  static const int v = [] { const char* e = getenv("NEURST_LN_BWD_SMALL_THREADS"); const int t = e ? atoi(e) : 1024; return (t == 256 || t == 512) ? t : 1024; }();
  return v;
}
static int ln_bwd_threads(int64_t rows) { return rows >= 4096 ? 1024 : (rows >= 512 ? ln_small_threads() : 256); }
extern "C" int64_t neurst_layernorm_bwd_blocks(int64_t rows) {
  const int64_t per = ln_bwd_threads(rows) / 64, cap = rows >= 4096 ? 256 : 512;
  const int64_t g = cdiv64(rows, per);
  return g > cap ? cap : (g < 1 ? 1 : g);
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
  const uint32_t inner0 = fmix32(key);
  const int idx32 = (double)rows * (double)C < 4294967296.0;
  // persistent waves: at most 1024 workgroups, each wave walks rows with stride (measured best: 1 row in flight per wave)
  const int gv_ = (int)(g > 1024 ? 1024 : g);
#define FWD_VEC(TT, VEC, NCH)                                                                                              \
  if (g_opt_ln_prefetch & 1)                                                                                               \
    ln_fwd_vec<TT, VEC, NCH, 1, true><<<gv_, 256, 0, s>>>((const TT*)x, (TT*)yz, gamma, beta, (TT*)out, mean, rstd, rows, (int)C, \
                                                          thr, sc, key, inner0, idx32, lens, T);                           \
  else                                                                                                                     \
    ln_fwd_vec<TT, VEC, NCH, 1, false><<<gv_, 256, 0, s>>>((const TT*)x, (TT*)yz, gamma, beta, (TT*)out, mean, rstd, rows, (int)C, \
                                                           thr, sc, key, inner0, idx32, lens, T)
#define FWD_DISPATCH(TT, VEC)                                         \
  switch (ln_nch(C, VEC)) {                                           \
    case 1: FWD_VEC(TT, VEC, 1); break;                               \
    case 2: FWD_VEC(TT, VEC, 2); break;                               \
    case 3: FWD_VEC(TT, VEC, 3); break;                               \
    case 4: FWD_VEC(TT, VEC, 4); break;                               \
    default: FWD_VEC(TT, VEC, 8); break;                              \
  }
  if (dtype == NEURST_BF16 && (g_opt_ln_prefetch & 16) && !lens && idx32 && ln_line_shape(C) &&
      ((((uintptr_t)x | (uintptr_t)yz | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0)) {
#define FWD_LINE(VEC, NCH)                                                                                                           \
    if (thr) ln_fwd_line<VEC, NCH, true><<<gv_, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)yz, gamma, beta, (bf16_t*)out, mean, rstd, rows, thr, sc, inner0); \
    else ln_fwd_line<VEC, NCH, false><<<gv_, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)yz, gamma, beta, (bf16_t*)out, mean, rstd, rows, thr, sc, inner0)
    switch ((int)C) {
      case 256: FWD_LINE(4, 1); break;
      case 512: FWD_LINE(8, 1); break;
      default: FWD_LINE(4, 3); break;
    }
#undef FWD_LINE
  } else if (dtype == NEURST_BF16) {
    if (ln_vec_ok(C, 8, x, yz, out)) { if (ln_lanes8(C, 4)) { FWD_DISPATCH(bf16_t, 4) } else { FWD_DISPATCH(bf16_t, 8) } }
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
                                                float* dbias, int64_t rows, int64_t C, float rate, uint32_t key, const int32_t* lens,
                                                int64_t T, int dtype, void* stream) {
  NEURST_CHECK_ARG(rows > 0 && C > 0 && rate >= 0.f && rate < 1.f, "layernorm_bwd: bad args");
  NEURST_CHECK_ARG(rate == 0.f || (dy && dy != dz), "layernorm_bwd: dropout needs a separate dy buffer");
  NEURST_CHECK_ARG(!lens || T > 0, "layernorm_bwd: lens given without T");
  uint32_t thr = rate > 0.f ? rate_threshold(rate) : 0u;
  float sc = keep_scale(rate);
  int grid = (int)neurst_layernorm_bwd_blocks(rows);
  hipStream_t s = as_stream(stream);
  const int nthr = ln_bwd_threads(rows);
  // dbias rides in the same kernel when its third [waves][C] LDS array still fits; otherwise a column sum of dy follows
  const bool fuse_db = dbias && (size_t)3 * (nthr / 64) * C * sizeof(float) <= 150 * 1024;
  size_t shmem = (size_t)(fuse_db ? 3 : 2) * (nthr / 64) * C * sizeof(float);
  bool vec = false;
  // column sums: workgroups add their LDS-reduced partials straight into dgamma / dbeta (measured 45 us against
  // 38 + 12 us for partial buffers plus a second reduction launch); `partial` is kept in the ABI and, when given,
  // selects the deterministic two-pass reduction
  const bool use_partial = partial != nullptr;
  const uint32_t inner0 = fmix32(key);
  const int idx32 = (double)rows * (double)C < 4294967296.0;
#define BWD_VEC_DB(TT, VEC, NCH, DB, PF)                                                                                     \
  if (shmem > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_vec<TT, VEC, NCH, DB, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
  ln_bwd_vec<TT, VEC, NCH, DB, PF><<<grid, nthr, shmem, s>>>((const TT*)dout, (const TT*)z, gamma, mean, rstd, (TT*)dz, (TT*)dy, use_partial ? partial : nullptr, \
                                                            dgamma, dbeta, dbias, rows, (int)C, thr, sc, key, inner0, idx32, lens, T)
#define BWD_VEC(TT, VEC, NCH)                                                                                                \
  if (g_opt_ln_prefetch & 2) { if (fuse_db) { BWD_VEC_DB(TT, VEC, NCH, true, true); } else { BWD_VEC_DB(TT, VEC, NCH, false, true); } } \
  else { if (fuse_db) { BWD_VEC_DB(TT, VEC, NCH, true, false); } else { BWD_VEC_DB(TT, VEC, NCH, false, false); } }
#define BWD_DISPATCH(TT, VEC)                                         \
  switch (ln_nch(C, VEC)) {                                           \
    case 1: BWD_VEC(TT, VEC, 1); break;                               \
    case 2: BWD_VEC(TT, VEC, 2); break;                               \
    case 3: BWD_VEC(TT, VEC, 3); break;                               \
    case 4: BWD_VEC(TT, VEC, 4); break;                               \
    default: BWD_VEC(TT, VEC, 8); break;                              \
  }
  if (dtype == NEURST_BF16 && (g_opt_ln_prefetch & 32) && !lens && idx32 && !use_partial && !dbias && ln_line_shape(C) &&
      ((((uintptr_t)dout | (uintptr_t)z | (uintptr_t)dz | (uintptr_t)gamma | (thr ? (uintptr_t)dy : 0)) & 15) == 0)) {
    const size_t lsm = (size_t)2 * (nthr / 64) * C * sizeof(float);
#define BWD_LINE(VEC, NCH)                                                                                                           \
    if (thr) {                                                                                                                       \
      if (lsm > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_line<VEC, NCH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lsm); \
      ln_bwd_line<VEC, NCH, true><<<grid, nthr, lsm, s>>>((const bf16_t*)dout, (const bf16_t*)z, gamma, mean, rstd, (bf16_t*)dz, (bf16_t*)dy, dgamma, dbeta, rows, thr, sc, inner0); \
    } else {                                                                                                                         \
      if (lsm > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_line<VEC, NCH, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lsm); \
      ln_bwd_line<VEC, NCH, false><<<grid, nthr, lsm, s>>>((const bf16_t*)dout, (const bf16_t*)z, gamma, mean, rstd, (bf16_t*)dz, (bf16_t*)dy, dgamma, dbeta, rows, thr, sc, inner0); \
    }
    switch ((int)C) {
      case 256: BWD_LINE(4, 1); break;
      case 512: BWD_LINE(8, 1); break;
      default: BWD_LINE(4, 3); break;
    }
#undef BWD_LINE
    NEURST_CHECK_LAUNCH("dropout_add_layernorm_bwd");
    return NEURST_OK;
  }
  if (dtype == NEURST_BF16) {
    if (ln_vec_ok(C, 8, dout, z, dz) && (!thr || ((uintptr_t)dy & 15) == 0) && shmem <= 150 * 1024) { vec = true; if (ln_lanes8(C, 8)) { BWD_DISPATCH(bf16_t, 4) } else { BWD_DISPATCH(bf16_t, 8) } }
    else ln_bwd_generic<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)dout, (const bf16_t*)z, gamma, mean, rstd, (bf16_t*)dz, (bf16_t*)dy, dgamma, dbeta, rows, (int)C, thr, sc, key, lens, T);
  } else {
    if (ln_vec_ok(C, 4, dout, z, dz) && (!thr || ((uintptr_t)dy & 15) == 0) && shmem <= 150 * 1024) { vec = true; BWD_DISPATCH(float, 4) }
    else ln_bwd_generic<float><<<grid, 256, 0, s>>>((const float*)dout, (const float*)z, gamma, mean, rstd, (float*)dz, (float*)dy, dgamma, dbeta, rows, (int)C, thr, sc, key, lens, T);
  }
#undef BWD_DISPATCH
#undef BWD_VEC
#undef BWD_VEC_DB
  NEURST_CHECK_LAUNCH("dropout_add_layernorm_bwd");
  if (dbias && !(vec && fuse_db)) {   // generic path, or a row too wide for the third LDS array
    const int rc = neurst_colsum(thr ? dy : dz, dbias, rows, C, C, dtype, 1, stream);
    if (rc != NEURST_OK) return rc;
  }
  if (vec && use_partial) {
    ln_bwd_reduce<<<(int)cdiv64(2 * C, 64), 1024, 0, s>>>(partial, dgamma, dbeta, grid, (int)C);
    NEURST_CHECK_LAUNCH("layernorm_bwd_reduce");
  }
  return NEURST_OK;
}

// ---------------------------------------------------------------- allocation-free variant (no reference counterpart)
// For callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid it):
// the deterministic two-pass column reduction of the backward above, with its [2][nblk][C] fp32 block partials in a
// file-scope device array instead of a caller workspace.  Every GPU holds its own copy of the array (4 MiB, until the
// process ends) at its own address: the address is looked up once per device with hipGetSymbolAddress and handed to the
// kernels as an ordinary pointer - the array's name is never used as an address in host code.  Every block slot that the
// reduction reads is written by the first pass of the same call, so nothing has to be cleared between calls; two calls
// that overlap on one GPU (different streams) share the array and race.
constexpr int64_t LN_SCRATCH_FLOATS = 1 << 20;                      // 4 MiB: e.g. 512 blocks x 2 x 1024 columns
static __device__ __align__(256) float scratch[LN_SCRATCH_FLOATS];
constexpr int LN_MAX_DEVICES = 64;
static float* g_scratch_ptr[LN_MAX_DEVICES];                        // per device, filled on that device's first call
#include <mutex>
static std::mutex g_scratch_mu;

static int ln_scratch_for_current_device(float** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= LN_MAX_DEVICES) {
    neurst_set_error("layernorm_bwd_noalloc: cannot identify the current device (%s)", hipGetErrorString(e));
    return NEURST_E_LAUNCH;
  }
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    float* p = nullptr;
    e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) {
      neurst_set_error("layernorm_bwd_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
      return NEURST_E_LAUNCH;
    }
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return NEURST_OK;
}

extern "C" int neurst_dropout_add_layernorm_bwd_noalloc(const void* dout, const void* z, const float* gamma, const float* mean,
                                                        const float* rstd, void* dz, void* dy, float* dgamma, float* dbeta,
                                                        float* dbias, int64_t rows, int64_t C, float rate, uint32_t key,
                                                        const int32_t* lens, int64_t T, int dtype, void* stream) {
  NEURST_CHECK_ARG(rows > 0 && C > 0, "layernorm_bwd_noalloc: bad shape");
  const int64_t need = 2 * neurst_layernorm_bwd_blocks(rows) * C;
  NEURST_CHECK_ARG(need <= LN_SCRATCH_FLOATS,
                   "layernorm_bwd_noalloc: rows=%ld x C=%ld needs %ld floats of block partials, the built-in scratch holds %ld "
                   "(use neurst_dropout_add_layernorm_bwd with a caller workspace)",
                   (long)rows, (long)C, (long)need, (long)LN_SCRATCH_FLOATS);
  float* ptr = nullptr;
  const int rc = ln_scratch_for_current_device(&ptr);
  if (rc != NEURST_OK) return rc;
  return neurst_dropout_add_layernorm_bwd(dout, z, gamma, mean, rstd, dz, dy, dgamma, dbeta, ptr, dbias, rows, C, rate, key, lens, T,
                                          dtype, stream);
}
