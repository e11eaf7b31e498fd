// K4/K4c/K4x: fused multi-head attention forward + backward (flash style, the
// [B,H,Tq,Tk] matrix is never materialised).  Reference arithmetic:
// st/models/tools/utils.py:545-553 (q *= dh^-0.5, split/merge heads) and
// st/layers/common_attention_v1.py:180-204,325-365 (additive -1e9 biases, softmax,
// dropout on the weights, weights.v).
//
// Layout: one workgroup = 4 waves = 64 query rows (forward, dQ) or 64 keys (dK/dV);
// K/V (or Q/dO) tiles of 64 rows are staged through LDS; every product runs on
// 16x16 MFMA tiles (bf16: 16x16x32, fp32 parity mode: 16x16x4); softmax statistics are
// wavefront shuffles over the 16 lanes that share a row.
#include "common.h"

#define NEG_BIAS (-1e9f)

template <typename T> struct ACfg;
template <> struct ACfg<float> { static constexpr int EPC = 4, PAD = 4; typedef f32x4 V; static constexpr bool TR = false; };
template <> struct ACfg<bf16_t> { static constexpr int EPC = 8, PAD = 8; typedef bf16x8 V; static constexpr bool TR = false; };

template <typename T> __device__ __forceinline__ float fast_exp(float x);
template <> __device__ __forceinline__ float fast_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ float fast_exp<bf16_t>(float x) { return __expf(x); }

template <typename T>
__device__ __forceinline__ typename ACfg<T>::V load_row_chunk(const T* __restrict__ row, int c, int dh, bool valid, bool aligned) {
  constexpr int EPC = ACfg<T>::EPC;
  typename ACfg<T>::V v;
  if (valid && aligned && c + EPC <= dh) {
    v = *reinterpret_cast<const typename ACfg<T>::V*>(row + c);
  } else {
#pragma unroll
    for (int j = 0; j < EPC; ++j) v[j] = (valid && c + j < dh) ? row[c + j] : from_f32<T>(0.f);
  }
  return v;
}

// S[r][c] = g[(row0+r)*ldg + c] for r<64, c<DH (zero outside nrows / dh)
template <typename T, int DH>
__device__ __forceinline__ void stage_natural(T* __restrict__ S, const T* __restrict__ g, int64_t ldg, int64_t row0, int64_t nrows, int dh,
                                              bool aligned, int tid) {
  constexpr int EPC = ACfg<T>::EPC, LD = DH + ACfg<T>::PAD, CPR = DH / EPC;
  for (int c = tid; c < 64 * CPR; c += 256) {
    int r = c / CPR, cc = (c % CPR) * EPC;
    auto v = load_row_chunk<T>(g + (row0 + r) * ldg, cc, dh, row0 + r < nrows, aligned);
    *reinterpret_cast<typename ACfg<T>::V*>(S + r * LD + cc) = v;
  }
}
// St[c][r] = g[(row0+r)*ldg + c]   (row index becomes the contiguous one)
template <typename T, int DH>
__device__ __forceinline__ void stage_transposed(T* __restrict__ St, const T* __restrict__ g, int64_t ldg, int64_t row0, int64_t nrows,
                                                 int dh, bool aligned, int tid) {
  constexpr int EPC = ACfg<T>::EPC, LDT = 64 + ACfg<T>::PAD, CPR = DH / EPC;
  for (int c = tid; c < 64 * CPR; c += 256) {
    int r = c / CPR, cc = (c % CPR) * EPC;
    auto v = load_row_chunk<T>(g + (row0 + r) * ldg, cc, dh, row0 + r < nrows, aligned);
    if constexpr (sizeof(T) == 2) {
      typedef unsigned short __attribute__((may_alias)) u16_alias;
      const u32x4 a = __builtin_bit_cast(u32x4, v);
#pragma unroll
      for (int j = 0; j < EPC; ++j)
        *reinterpret_cast<u16_alias*>(St + (cc + j) * LDT + r) = (unsigned short)((a[j >> 1] >> (16 * (j & 1))) & 0xFFFFu);
    } else {
#pragma unroll
      for (int j = 0; j < EPC; ++j) St[(cc + j) * LDT + r] = v[j];
    }
  }
}

// A-operand fragments of 16 rows straight from global memory: frag[ks] covers k in [ks*KS, ks*KS+KS)
template <typename T, int DH>
__device__ __forceinline__ void load_frags_global(typename Mma<T>::Frag (&f)[DH / Mma<T>::KS], const T* __restrict__ g, int64_t ldg,
                                                  int64_t row0, int64_t nrows, int dh, bool aligned, int lane) {
  constexpr int KS = Mma<T>::KS;
  const int64_t row = row0 + (lane & 15);
  const bool valid = row < nrows;
  const T* p = g + row * ldg;
#pragma unroll
  for (int ks = 0; ks < DH / KS; ++ks) {
    if constexpr (sizeof(T) == 4) {
      int c = ks * KS + (lane >> 4);
      f[ks] = (valid && c < dh) ? p[c] : 0.f;
    } else {
      f[ks] = load_row_chunk<T>(p, ks * KS + 8 * (lane >> 4), dh, valid, aligned);
    }
  }
}

// B-operand fragment whose k index runs over the ROWS of a natural [row][col] LDS tile: bf16 uses the
// transposing read ds_read_b64_tr_b16, fp32 a plain strided read.  (tr / ld_tr are unused, kept for call sites.)
template <typename T>
__device__ __forceinline__ typename Mma<T>::Frag load_b_rowk(const T* nat, int ld_nat, const T* tr, int ld_tr, int k0, int col0, int lane) {
  return Mma<T>::load_tr(nat + k0 * ld_nat + col0, ld_nat, lane);
}

// wave-private LDS hand-off: LDS operations of one wave execute in order, only the compiler must not reorder
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct AttnArgs {
  const void *q, *k, *v, *o, *d_o;
  void *out, *dq, *dk, *dv;
  float* lse; float* delta;
  const float* lse_in;
  int64_t ldq, ldk, ldv, ldo, lddq, lddk, lddv;
  int64_t bsq, bsk, bsv, bso;   // batch strides (elements) of q / k / v / o
  int64_t B, H, Tq, Tk;
  int dh, mask_mode;
  const int32_t* len_k;
  uint32_t thr, key, inner0;   // inner0 = fmix32(key): the inner hash when every flat index fits 32 bits
  int idx32;
  float keep_scale, qscale;
  int q_aligned, k_aligned, v_aligned, o_aligned;
};

// dropout decision for weight (row, col): row_base = flat index of the row's first weight
__device__ __forceinline__ bool attn_keep(const AttnArgs& a, uint64_t row_base, int col) {
  if (a.idx32) return (fmix32(((uint32_t)row_base + (uint32_t)col) ^ a.inner0) >> 8) >= a.thr;
  return dropout_keep(a.key, row_base + (uint64_t)col, a.thr);
}

__device__ __forceinline__ float group16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64)); v = fmaxf(v, __shfl_xor(v, 8, 64));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}

// score of (query qi, key kj) after scaling and the reference's additive biases; -inf outside the tensor
__device__ __forceinline__ float biased_score(float raw, float qscale, int64_t qi, int64_t kj, int64_t Tk, int mask_mode, int len) {
  if (kj >= Tk) return -INFINITY;
  float s = raw * qscale;
  if (mask_mode == NEURST_MASK_KEYPAD) { if (kj >= len) s += NEG_BIAS; }
  else if (mask_mode == NEURST_MASK_CAUSAL) { if (kj > qi) s += NEG_BIAS; }
  return s;
}

// ================================================================== forward
template <typename T, int DH>
__global__ void __launch_bounds__(256) mha_fwd_kernel(AttnArgs a) {
  constexpr int KS = Mma<T>::KS, LDK = DH + ACfg<T>::PAD, LDT = 64 + ACfg<T>::PAD, LDP = 64 + ACfg<T>::PAD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Ks = reinterpret_cast<T*>(smem);            // [64][LDK]
  T* Vs = Ks + 64 * LDK;                         // fp32: [64][LDK] natural; bf16: [DH][LDT] transposed
  T* Ps = Vs + (ACfg<T>::TR ? DH * LDT : 64 * LDK);  // [4][16][LDP]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t b = blockIdx.z, h = blockIdx.y, q0 = (int64_t)blockIdx.x * 64;
  const int dh = a.dh;
  const T* qg = reinterpret_cast<const T*>(a.q) + b * a.bsq + h * dh;
  const T* kg = reinterpret_cast<const T*>(a.k) + b * a.bsk + h * dh;
  const T* vg = reinterpret_cast<const T*>(a.v) + b * a.bsv + h * dh;
  const int len = a.mask_mode == NEURST_MASK_KEYPAD ? a.len_k[b] : 0;
  int64_t kv_end = a.Tk;
  if (a.mask_mode == NEURST_MASK_CAUSAL) kv_end = q0 + 64 < a.Tk ? q0 + 64 : a.Tk;
  else if (a.mask_mode == NEURST_MASK_KEYPAD && len > 0 && len < a.Tk) kv_end = len;

  typename Mma<T>::Frag qf[DH / KS];
  load_frags_global<T, DH>(qf, qg, a.ldq, q0 + w * 16, a.Tq, dh, a.q_aligned, lane);

  f32x4 oacc[DH / 16];
#pragma unroll
  for (int i = 0; i < DH / 16; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }
  T* Pw = Ps + w * 16 * LDP;
  const int64_t bh = b * a.H + h;
  const int64_t kv_lim = a.mask_mode == NEURST_MASK_KEYPAD ? (len < a.Tk ? len : a.Tk) : a.Tk;
  uint64_t rowb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) rowb[r] = (uint64_t)((bh * a.Tq + q0 + w * 16 + (lane >> 4) * 4 + r) * a.Tk);

  for (int64_t kv0 = 0; kv0 < kv_end; kv0 += 64) {
    __syncthreads();
    stage_natural<T, DH>(Ks, kg, a.ldk, kv0, a.Tk, dh, a.k_aligned, tid);
    if constexpr (ACfg<T>::TR) stage_transposed<T, DH>(Vs, vg, a.ldv, kv0, a.Tk, dh, a.v_aligned, tid);
    else stage_natural<T, DH>(Vs, vg, a.ldv, kv0, a.Tk, dh, a.v_aligned, tid);
    __syncthreads();
    f32x4 s[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      s[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DH / KS; ++ks)
        s[nt] = Mma<T>::mma(qf[ks], Mma<T>::load(Ks + nt * 16 * LDK + ks * KS, LDK, lane), s[nt]);
    }
    float mloc[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    const bool edge = a.mask_mode == NEURST_MASK_CAUSAL || kv0 + 64 > kv_lim;   // wave-uniform: tile touches a bias or the tensor end
    if (edge) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int64_t qi = q0 + w * 16 + (lane >> 4) * 4 + r, kj = kv0 + nt * 16 + (lane & 15);
          float v = biased_score(s[nt][r], a.qscale, qi, kj, a.Tk, a.mask_mode, len);
          s[nt][r] = v;
          mloc[r] = fmaxf(mloc[r], v);
        }
    } else {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = s[nt][r] * a.qscale;
          s[nt][r] = v;
          mloc[r] = fmaxf(mloc[r], v);
        }
    }
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mn = fmaxf(m_run[r], group16_max(mloc[r]));
      alpha[r] = (m_run[r] == -INFINITY) ? 0.f : fast_exp<T>(m_run[r] - mn);
      m_run[r] = mn;
      l_run[r] *= alpha[r];
    }
#pragma unroll
    for (int i = 0; i < DH / 16; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[i][r] *= alpha[r];
    const int kcol = (int)kv0 + (lane & 15);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = (edge && s[nt][r] == -INFINITY) ? 0.f : fast_exp<T>(s[nt][r] - m_run[r]);
        l_run[r] += p;
        if (a.thr) p = attn_keep(a, rowb[r], kcol + nt * 16) ? p * a.keep_scale : 0.f;
        Pw[((lane >> 4) * 4 + r) * LDP + nt * 16 + (lane & 15)] = from_f32<T>(p);
      }
    wave_lds_sync();
#pragma unroll
    for (int ks2 = 0; ks2 < 64 / KS; ++ks2) {
      typename Mma<T>::Frag pf = Mma<T>::load(Pw + ks2 * KS, LDP, lane);
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt)
        oacc[dt] = Mma<T>::mma(pf, load_b_rowk<T>(Vs, LDK, Vs, LDT, ks2 * KS, dt * 16, lane), oacc[dt]);
    }
  }
  T* og = reinterpret_cast<T*>(a.out) + b * a.bso + h * dh;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float l = group16_sum(l_run[r]);
    const float inv = 1.f / l;
    const int64_t qi = q0 + w * 16 + (lane >> 4) * 4 + r;
    if (qi < a.Tq) {
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        int c = dt * 16 + (lane & 15);
        if (c < dh) og[qi * a.ldo + c] = from_f32<T>(oacc[dt][r] * inv);
      }
      if ((lane & 15) == 0) a.lse[bh * a.Tq + qi] = m_run[r] + logf(l);
    }
  }
}

// ================================================================== delta = rowsum(dO * O)
template <typename T>
__global__ void mha_delta_kernel(const T* __restrict__ o, const T* __restrict__ d_o, float* __restrict__ delta, int64_t ldo, int64_t B,
                                 int64_t H, int64_t Tq, int dh) {
  const int lane = threadIdx.x & 63;
  const int64_t total = B * H * Tq;
  for (int64_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < total; i += (int64_t)gridDim.x * 4) {
    int64_t qi = i % Tq, bh = i / Tq;
    int64_t h = bh % H, b = bh / H;
    const int64_t off = (b * Tq + qi) * ldo + h * dh;
    float s = 0.f;
    for (int c = lane; c < dh; c += 64) s += to_f32<T>(o[off + c]) * to_f32<T>(d_o[off + c]);
    s = wave_sum(s);
    if (lane == 0) delta[i] = s;
  }
}

// ================================================================== backward: dQ
template <typename T, int DH>
__global__ void __launch_bounds__(256) mha_bwd_dq_kernel(AttnArgs a) {
  constexpr int KS = Mma<T>::KS, LDK = DH + ACfg<T>::PAD, LDT = 64 + ACfg<T>::PAD, LDP = 64 + ACfg<T>::PAD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Ks = reinterpret_cast<T*>(smem);             // [64][LDK]
  T* Vs = Ks + 64 * LDK;                          // [64][LDK]
  T* Kt = Vs + 64 * LDK;                          // bf16 only: [DH][LDT]
  T* Ps = Kt + (ACfg<T>::TR ? DH * LDT : 0);      // [4][16][LDP]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t b = blockIdx.z, h = blockIdx.y, q0 = (int64_t)blockIdx.x * 64;
  const int dh = a.dh;
  const T* qg = reinterpret_cast<const T*>(a.q) + b * a.bsq + h * dh;
  const T* kg = reinterpret_cast<const T*>(a.k) + b * a.bsk + h * dh;
  const T* vg = reinterpret_cast<const T*>(a.v) + b * a.bsv + h * dh;
  const T* dog = reinterpret_cast<const T*>(a.d_o) + b * a.bso + h * dh;
  const int len = a.mask_mode == NEURST_MASK_KEYPAD ? a.len_k[b] : 0;
  int64_t kv_end = a.Tk;
  if (a.mask_mode == NEURST_MASK_CAUSAL) kv_end = q0 + 64 < a.Tk ? q0 + 64 : a.Tk;
  else if (a.mask_mode == NEURST_MASK_KEYPAD && len > 0 && len < a.Tk) kv_end = len;
  const int64_t bh = b * a.H + h;

  typename Mma<T>::Frag qf[DH / KS], dof[DH / KS];
  load_frags_global<T, DH>(qf, qg, a.ldq, q0 + w * 16, a.Tq, dh, a.q_aligned, lane);
  load_frags_global<T, DH>(dof, dog, a.ldo, q0 + w * 16, a.Tq, dh, a.o_aligned, lane);
  float lse_r[4], del_r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int64_t qi = q0 + w * 16 + (lane >> 4) * 4 + r;
    lse_r[r] = qi < a.Tq ? a.lse_in[bh * a.Tq + qi] : 0.f;
    del_r[r] = qi < a.Tq ? a.delta[bh * a.Tq + qi] : 0.f;
  }
  f32x4 dqacc[DH / 16];
#pragma unroll
  for (int i = 0; i < DH / 16; ++i) dqacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  T* Pw = Ps + w * 16 * LDP;
  const int64_t kv_lim = a.mask_mode == NEURST_MASK_KEYPAD ? (len < a.Tk ? len : a.Tk) : a.Tk;
  const bool q_tail = q0 + 64 > a.Tq;
  uint64_t rowb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) rowb[r] = (uint64_t)((bh * a.Tq + q0 + w * 16 + (lane >> 4) * 4 + r) * a.Tk);

  for (int64_t kv0 = 0; kv0 < kv_end; kv0 += 64) {
    __syncthreads();
    stage_natural<T, DH>(Ks, kg, a.ldk, kv0, a.Tk, dh, a.k_aligned, tid);
    stage_natural<T, DH>(Vs, vg, a.ldv, kv0, a.Tk, dh, a.v_aligned, tid);
    if constexpr (ACfg<T>::TR) stage_transposed<T, DH>(Kt, kg, a.ldk, kv0, a.Tk, dh, a.k_aligned, tid);
    __syncthreads();
    const bool edge = a.mask_mode == NEURST_MASK_CAUSAL || kv0 + 64 > kv_lim || q_tail;
    const int kcol = (int)kv0 + (lane & 15);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DH / KS; ++ks) {
        s = Mma<T>::mma(qf[ks], Mma<T>::load(Ks + nt * 16 * LDK + ks * KS, LDK, lane), s);
        dp = Mma<T>::mma(dof[ks], Mma<T>::load(Vs + nt * 16 * LDK + ks * KS, LDK, lane), dp);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p;
        if (edge) {
          int64_t qi = q0 + w * 16 + (lane >> 4) * 4 + r, kj = kv0 + nt * 16 + (lane & 15);
          float sc = biased_score(s[r], a.qscale, qi, kj, a.Tk, a.mask_mode, len);
          p = (sc == -INFINITY || qi >= a.Tq) ? 0.f : fast_exp<T>(sc - lse_r[r]);
        } else {
          p = fast_exp<T>(s[r] * a.qscale - lse_r[r]);
        }
        float dpe = dp[r];
        if (a.thr) dpe = attn_keep(a, rowb[r], kcol + nt * 16) ? dpe * a.keep_scale : 0.f;
        Pw[((lane >> 4) * 4 + r) * LDP + nt * 16 + (lane & 15)] = from_f32<T>(p * (dpe - del_r[r]));
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int ks2 = 0; ks2 < 64 / KS; ++ks2) {
      typename Mma<T>::Frag pf = Mma<T>::load(Pw + ks2 * KS, LDP, lane);
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt)
        dqacc[dt] = Mma<T>::mma(pf, load_b_rowk<T>(Ks, LDK, Kt, LDT, ks2 * KS, dt * 16, lane), dqacc[dt]);
    }
  }
  T* dqg = reinterpret_cast<T*>(a.dq) + b * a.Tq * a.lddq + h * dh;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t qi = q0 + w * 16 + (lane >> 4) * 4 + r;
    if (qi < a.Tq) {
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        int c = dt * 16 + (lane & 15);
        if (c < dh) dqg[qi * a.lddq + c] = from_f32<T>(dqacc[dt][r] * a.qscale);
      }
    }
  }
}

// ================================================================== backward: dK, dV
template <typename T, int DH>
__global__ void __launch_bounds__(256) mha_bwd_dkv_kernel(AttnArgs a) {
  constexpr int KS = Mma<T>::KS, LDK = DH + ACfg<T>::PAD, LDT = 64 + ACfg<T>::PAD, LDP = 64 + ACfg<T>::PAD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Qs = reinterpret_cast<T*>(smem);                 // [64][LDK]
  T* dOs = Qs + 64 * LDK;                             // [64][LDK]
  T* Qt = dOs + 64 * LDK;                             // bf16 only [DH][LDT]
  T* dOt = Qt + (ACfg<T>::TR ? DH * LDT : 0);         // bf16 only [DH][LDT]
  T* Ps = dOt + (ACfg<T>::TR ? DH * LDT : 0);         // [4][16][LDP]  dropped P^T
  T* dSs = Ps + 4 * 16 * LDP;                         // [4][16][LDP]  dS^T
  float* lse_s = reinterpret_cast<float*>(dSs + 4 * 16 * LDP);  // [64]
  float* del_s = lse_s + 64;                          // [64]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t b = blockIdx.z, h = blockIdx.y, kv0 = (int64_t)blockIdx.x * 64;
  const int dh = a.dh;
  const T* qg = reinterpret_cast<const T*>(a.q) + b * a.bsq + h * dh;
  const T* kg = reinterpret_cast<const T*>(a.k) + b * a.bsk + h * dh;
  const T* vg = reinterpret_cast<const T*>(a.v) + b * a.bsv + h * dh;
  const T* dog = reinterpret_cast<const T*>(a.d_o) + b * a.bso + h * dh;
  const int len = a.mask_mode == NEURST_MASK_KEYPAD ? a.len_k[b] : 0;
  const int64_t bh = b * a.H + h;
  const bool dead = a.mask_mode == NEURST_MASK_KEYPAD && len > 0 && kv0 >= len;  // padded keys get exactly zero weight

  typename Mma<T>::Frag kf[DH / KS], vf[DH / KS];
  load_frags_global<T, DH>(kf, kg, a.ldk, kv0 + w * 16, a.Tk, dh, a.k_aligned, lane);
  load_frags_global<T, DH>(vf, vg, a.ldv, kv0 + w * 16, a.Tk, dh, a.v_aligned, lane);
  f32x4 dkacc[DH / 16], dvacc[DH / 16];
#pragma unroll
  for (int i = 0; i < DH / 16; ++i) { dkacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  T* Pw = Ps + w * 16 * LDP;
  T* dSw = dSs + w * 16 * LDP;
  int64_t q_begin = 0;
  if (a.mask_mode == NEURST_MASK_CAUSAL) q_begin = kv0;  // queries before the first key of this tile are fully masked
  if (dead) q_begin = a.Tq;
  const int64_t kv_lim = a.mask_mode == NEURST_MASK_KEYPAD ? (len < a.Tk ? len : a.Tk) : a.Tk;

  for (int64_t q0 = q_begin; q0 < a.Tq; q0 += 64) {
    __syncthreads();
    stage_natural<T, DH>(Qs, qg, a.ldq, q0, a.Tq, dh, a.q_aligned, tid);
    stage_natural<T, DH>(dOs, dog, a.ldo, q0, a.Tq, dh, a.o_aligned, tid);
    if constexpr (ACfg<T>::TR) {
      stage_transposed<T, DH>(Qt, qg, a.ldq, q0, a.Tq, dh, a.q_aligned, tid);
      stage_transposed<T, DH>(dOt, dog, a.ldo, q0, a.Tq, dh, a.o_aligned, tid);
    }
    if (tid < 64) {
      int64_t qi = q0 + tid;
      lse_s[tid] = qi < a.Tq ? a.lse_in[bh * a.Tq + qi] : 0.f;
      del_s[tid] = qi < a.Tq ? a.delta[bh * a.Tq + qi] : 0.f;
    }
    __syncthreads();
    const bool edge = a.mask_mode == NEURST_MASK_CAUSAL || kv0 + 64 > kv_lim || q0 + 64 > a.Tq;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DH / KS; ++ks) {
        st = Mma<T>::mma(kf[ks], Mma<T>::load(Qs + nt * 16 * LDK + ks * KS, LDK, lane), st);
        dpt = Mma<T>::mma(vf[ks], Mma<T>::load(dOs + nt * 16 * LDK + ks * KS, LDK, lane), dpt);
      }
      const int qloc = nt * 16 + (lane & 15);
      const int64_t qi = q0 + qloc;
      const float lse_q = lse_s[qloc], del_q = del_s[qloc];
      const uint64_t rowb = (uint64_t)((bh * a.Tq + qi) * a.Tk);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kcol = (int)kv0 + w * 16 + (lane >> 4) * 4 + r;
        float p;
        if (edge) {
          float sc = biased_score(st[r], a.qscale, qi, (int64_t)kcol, a.Tk, a.mask_mode, len);
          p = (sc == -INFINITY || qi >= a.Tq) ? 0.f : fast_exp<T>(sc - lse_q);
        } else {
          p = fast_exp<T>(st[r] * a.qscale - lse_q);
        }
        float pd = p, dpe = dpt[r];
        if (a.thr) {
          bool keep = attn_keep(a, rowb, kcol);
          pd = keep ? p * a.keep_scale : 0.f;
          dpe = keep ? dpe * a.keep_scale : 0.f;
        }
        Pw[((lane >> 4) * 4 + r) * LDP + qloc] = from_f32<T>(pd);
        dSw[((lane >> 4) * 4 + r) * LDP + qloc] = from_f32<T>(p * (dpe - del_q));
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int ks2 = 0; ks2 < 64 / KS; ++ks2) {
      typename Mma<T>::Frag pf = Mma<T>::load(Pw + ks2 * KS, LDP, lane);
      typename Mma<T>::Frag dsf = Mma<T>::load(dSw + ks2 * KS, LDP, lane);
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        dvacc[dt] = Mma<T>::mma(pf, load_b_rowk<T>(dOs, LDK, dOt, LDT, ks2 * KS, dt * 16, lane), dvacc[dt]);
        dkacc[dt] = Mma<T>::mma(dsf, load_b_rowk<T>(Qs, LDK, Qt, LDT, ks2 * KS, dt * 16, lane), dkacc[dt]);
      }
    }
  }
  T* dkg = reinterpret_cast<T*>(a.dk) + b * a.Tk * a.lddk + h * dh;
  T* dvg = reinterpret_cast<T*>(a.dv) + b * a.Tk * a.lddv + h * dh;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t kj = kv0 + w * 16 + (lane >> 4) * 4 + r;
    if (kj < a.Tk) {
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        int c = dt * 16 + (lane & 15);
        if (c < dh) {
          dkg[kj * a.lddk + c] = from_f32<T>(dkacc[dt][r] * a.qscale);
          dvg[kj * a.lddv + c] = from_f32<T>(dvacc[dt][r]);
        }
      }
    }
  }
}

// ================================================================== host side
template <typename T, int DH> static size_t fwd_smem() {
  constexpr int LDK = DH + ACfg<T>::PAD, LDT = 64 + ACfg<T>::PAD, LDP = 64 + ACfg<T>::PAD;
  return sizeof(T) * (64 * LDK + (ACfg<T>::TR ? DH * LDT : 64 * LDK) + 4 * 16 * LDP);
}
template <typename T, int DH> static size_t dq_smem() {
  constexpr int LDK = DH + ACfg<T>::PAD, LDT = 64 + ACfg<T>::PAD, LDP = 64 + ACfg<T>::PAD;
  return sizeof(T) * (2 * 64 * LDK + (ACfg<T>::TR ? DH * LDT : 0) + 4 * 16 * LDP);
}
template <typename T, int DH> static size_t dkv_smem() {
  constexpr int LDK = DH + ACfg<T>::PAD, LDT = 64 + ACfg<T>::PAD, LDP = 64 + ACfg<T>::PAD;
  return sizeof(T) * (2 * 64 * LDK + (ACfg<T>::TR ? 2 * DH * LDT : 0) + 8 * 16 * LDP) + 128 * sizeof(float);
}

template <typename K>
static int launch_attn(K kernel, dim3 grid, size_t smem, const AttnArgs& a, hipStream_t s, const char* name) {
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { neurst_set_error("%s: cannot reserve %zu B of LDS: %s", name, smem, hipGetErrorString(e)); return NEURST_E_LAUNCH; }
  }
  kernel<<<grid, 256, smem, s>>>(a);
  NEURST_CHECK_LAUNCH(name);
  return NEURST_OK;
}

static int fill_args(AttnArgs& a, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh, int mask_mode, const int32_t* len_k,
                     float rate, uint32_t key, int dtype) {
  NEURST_CHECK_ARG(B > 0 && H > 0 && Tq > 0 && Tk > 0 && dh > 0 && dh <= 192, "mha: bad shape (dh must be <= 192)");
  NEURST_CHECK_ARG(mask_mode >= 0 && mask_mode <= 2, "mha: bad mask mode");
  NEURST_CHECK_ARG(mask_mode != NEURST_MASK_KEYPAD || len_k, "mha: KEYPAD needs len_k");
  NEURST_CHECK_ARG(rate >= 0.f && rate < 1.f, "mha: bad dropout rate");
  NEURST_CHECK_ARG(B < 65536 && H < 65536, "mha: B and H must be < 65536");
  if (!a.bsq) a.bsq = Tq * a.ldq;
  if (!a.bsk) a.bsk = Tk * a.ldk;
  if (!a.bsv) a.bsv = Tk * a.ldv;
  if (!a.bso) a.bso = Tq * a.ldo;
  a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.dh = (int)dh; a.mask_mode = mask_mode; a.len_k = len_k;
  a.thr = rate > 0.f ? rate_threshold(rate) : 0u;
  a.key = key; a.keep_scale = keep_scale(rate);
  a.inner0 = fmix32(key);   // == fmix32(hi ^ key) with hi == 0
  a.idx32 = (double)B * (double)H * (double)Tq * (double)Tk < 4294967296.0;
  a.qscale = 1.0f / sqrtf((float)dh);
  size_t es = dtype_size(dtype);
  auto al = [&](const void* p, int64_t ld) { return (int)((((uintptr_t)p & 15) == 0) && ((ld * es) % 16 == 0) && ((dh * es) % 16 == 0)); };
  a.q_aligned = al(a.q, a.ldq); a.k_aligned = al(a.k, a.ldk); a.v_aligned = al(a.v, a.ldv);
  a.o_aligned = a.d_o ? al(a.d_o, a.ldo) : 1;
  return NEURST_OK;
}

#define DISPATCH_DH(T, KERNEL, SMEMFN, GRID, NAME)                                                       \
  do {                                                                                                   \
    if (dh <= 32) rc = launch_attn(KERNEL<T, 32>, GRID, SMEMFN<T, 32>(), a, s, NAME);                    \
    else if (dh <= 64) rc = launch_attn(KERNEL<T, 64>, GRID, SMEMFN<T, 64>(), a, s, NAME);               \
    else if (dh <= 96) rc = launch_attn(KERNEL<T, 96>, GRID, SMEMFN<T, 96>(), a, s, NAME);               \
    else if (dh <= 128) rc = launch_attn(KERNEL<T, 128>, GRID, SMEMFN<T, 128>(), a, s, NAME);            \
    else rc = launch_attn(KERNEL<T, 192>, GRID, SMEMFN<T, 192>(), a, s, NAME);                           \
  } while (0)

extern "C" int neurst_mha_fwd_strided(const void* q, const void* k, const void* v, void* o, float* lse, int64_t ldq, int64_t ldk,
                                      int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso, int64_t B, int64_t H,
                                      int64_t Tq, int64_t Tk, int64_t dh, int mask_mode, const int32_t* len_k, float rate, uint32_t key,
                                      int dtype, void* stream) {
  NEURST_CHECK_ARG(q && k && v && o && lse, "mha_fwd: null pointer");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.out = o; a.lse = lse; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.bsq = bsq; a.bsk = bsk; a.bsv = bsv; a.bso = bso;
  int rc = fill_args(a, B, H, Tq, Tk, dh, mask_mode, len_k, rate, key, dtype);
  if (rc) return rc;
  NEURST_CHECK_ARG(ldq >= H * dh && ldk >= H * dh && ldv >= H * dh && ldo >= H * dh, "mha_fwd: row stride smaller than H*dh");
  size_t es = dtype_size(dtype);
  if ((a.bsq * es) % 16) a.q_aligned = 0;
  if ((a.bsk * es) % 16) a.k_aligned = 0;
  if ((a.bsv * es) % 16) a.v_aligned = 0;
  hipStream_t s = as_stream(stream);
  dim3 grid((unsigned)cdiv64(Tq, 64), (unsigned)H, (unsigned)B);
  if (dtype == NEURST_BF16) DISPATCH_DH(bf16_t, mha_fwd_kernel, fwd_smem, grid, "mha_fwd");
  else DISPATCH_DH(float, mha_fwd_kernel, fwd_smem, grid, "mha_fwd");
  return rc;
}
extern "C" int neurst_mha_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int64_t ldq, int64_t ldk, int64_t ldv,
                              int64_t ldo, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh, int mask_mode,
                              const int32_t* len_k, float rate, uint32_t key, int dtype, void* stream) {
  return neurst_mha_fwd_strided(q, k, v, o, lse, ldq, ldk, ldv, ldo, Tq * ldq, Tk * ldk, Tk * ldv, Tq * ldo, B, H, Tq, Tk, dh, mask_mode,
                                len_k, rate, key, dtype, stream);
}

extern "C" int neurst_mha_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                              float* delta, void* dq, void* dk, void* dv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                              int64_t lddq, int64_t lddk, int64_t lddv, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh,
                              int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype, void* stream) {
  NEURST_CHECK_ARG(q && k && v && o && d_o && lse && delta && dq && dk && dv, "mha_bwd: null pointer");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.o = o; a.d_o = d_o; a.lse_in = lse; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  int rc = fill_args(a, B, H, Tq, Tk, dh, mask_mode, len_k, rate, key, dtype);
  if (rc) return rc;
  NEURST_CHECK_ARG(ldq >= H * dh && ldk >= H * dh && ldv >= H * dh && ldo >= H * dh && lddq >= H * dh && lddk >= H * dh && lddv >= H * dh,
                   "mha_bwd: row stride smaller than H*dh");
  hipStream_t s = as_stream(stream);
  int64_t nrows = B * H * Tq;
  int dgrid = (int)(cdiv64(nrows, 4) > 8192 ? 8192 : cdiv64(nrows, 4));
  if (dtype == NEURST_BF16) mha_delta_kernel<bf16_t><<<dgrid, 256, 0, s>>>((const bf16_t*)o, (const bf16_t*)d_o, delta, ldo, B, H, Tq, (int)dh);
  else mha_delta_kernel<float><<<dgrid, 256, 0, s>>>((const float*)o, (const float*)d_o, delta, ldo, B, H, Tq, (int)dh);
  NEURST_CHECK_LAUNCH("mha_delta");
  dim3 gq((unsigned)cdiv64(Tq, 64), (unsigned)H, (unsigned)B);
  dim3 gk((unsigned)cdiv64(Tk, 64), (unsigned)H, (unsigned)B);
  if (dtype == NEURST_BF16) DISPATCH_DH(bf16_t, mha_bwd_dq_kernel, dq_smem, gq, "mha_bwd_dq");
  else DISPATCH_DH(float, mha_bwd_dq_kernel, dq_smem, gq, "mha_bwd_dq");
  if (rc) return rc;
  if (dtype == NEURST_BF16) DISPATCH_DH(bf16_t, mha_bwd_dkv_kernel, dkv_smem, gk, "mha_bwd_dkv");
  else DISPATCH_DH(float, mha_bwd_dkv_kernel, dkv_smem, gk, "mha_bwd_dkv");
  return rc;
}
