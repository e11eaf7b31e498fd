// Shared device/host helpers for libneurst_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/neurst_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define NEURST_WAVE 64

// ---- error plumbing --------------------------------------------------------
void neurst_set_error(const char* fmt, ...);

#define NEURST_CHECK_ARG(cond, ...)                  \
  do {                                               \
    if (!(cond)) {                                   \
      neurst_set_error(__VA_ARGS__);                 \
      return NEURST_E_INVALID;                       \
    }                                                \
  } while (0)

#define NEURST_CHECK_LAUNCH(name)                                              \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      neurst_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return NEURST_E_LAUNCH;                                                  \
    }                                                                          \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t dtype_size(int dt) { return dt == NEURST_BF16 ? 2 : 4; }

// ---- scalar conversions ------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// ---- counter-based dropout RNG (same hash as oracle/ops.py:fmix32/keep_mask) --
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
// keep decision for flat element index idx under a call-site key
__device__ __forceinline__ bool dropout_keep(uint32_t key, uint64_t idx, uint32_t thr) {
  uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
  uint32_t h = fmix32(lo ^ fmix32(hi ^ key));
  return (h >> 8) >= thr;
}
static inline uint32_t rate_threshold(float rate) { return (uint32_t)(rate * 16777216.0f); }
// ---- attention-weight dropout: 32 decisions per word (oracle/ops.py:attn_keep_words - the rule is stated there) ----------
// widx = row * nw + w; W[k] = xorshift32^(k >> 2)(fmix32-hash of counter 4 widx + (k & 3)); bit i of the result = keep
// decision of key 32 w + i: the 16-bit number with digits NOT bit_i(W[k]) is below t16 = round((1 - rate) * 65536).
// inner0 = fmix32(key), idx32 = every counter fits 32 bits.  ~165 vector instructions per word (5 per decision; the
// per-weight hash it replaces: 19, two of them quarter-rate multiplies).
__host__ __device__ __forceinline__ uint32_t xorshift32(uint32_t x) {
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  return x;
}
static inline uint32_t attn_threshold16(float rate) {
  int t = (int)floor((1.0 - (double)rate) * 65536.0 + 0.5);
  return (uint32_t)(t < 1 ? 1 : (t > 65535 ? 65535 : t));
}
__device__ __forceinline__ uint32_t attn_keep_word(uint32_t key, uint32_t inner0, int idx32, uint64_t widx, uint32_t t16) {
  uint32_t c[4];
  if (idx32) {
    const uint32_t base = (uint32_t)widx * 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = fmix32((base + j) ^ inner0);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t ctr = widx * 4u + j;
      c[j] = fmix32((uint32_t)ctr ^ fmix32((uint32_t)(ctr >> 32) ^ key));
    }
  }
  uint32_t lt = 0u;
  // the threshold as a per-lane value on purpose: as a wave-uniform one the 16 digit tests become 16 SGPR-pair masks that
  // live across the caller's whole loop (measured in the attention forward: 56 spilled SGPRs, 177 v_readlane in the loop)
  uint32_t tv = t16;
  asm volatile("" : "+v"(tv));
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (k && (k & 3) == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] = xorshift32(c[j]);
    }
    const uint32_t W = c[k & 3], m = (uint32_t)__builtin_amdgcn_sbfe((int)tv, k, 1);   // all-ones / zero from digit k
    const uint32_t x = lt ^ W;
    lt = (x & m) | (~x & lt);                                     // digit set: lt | W, clear: lt & W
  }
  return lt;
}
static inline float keep_scale(float rate) { return 1.0f / (1.0f - rate); }

// ---- wave / block reductions ---------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// Same sum through the DPP cross-lane network instead of six LDS-crossbar shuffles (~60 instead of ~600 cycles of
// latency per reduction - what a one-row-per-wave kernel waits for twice per row).  Needs all 64 lanes active; the
// result is wave-uniform.  Steps: quad swaps, half-row mirror, row mirror (every lane of a 16-lane row now holds the
// row's sum), lane 15 -> next row (rows 1, 3), lane 31 -> rows 2, 3; lane 63 then holds the total.
#define NEURST_DPP_F32(x, ctrl, rows) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), (rows), 0xF, false))
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += NEURST_DPP_F32(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
  v += NEURST_DPP_F32(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
  v += NEURST_DPP_F32(v, 0x141, 0xF);   // row_half_mirror
  v += NEURST_DPP_F32(v, 0x140, 0xF);   // row_mirror
  v += NEURST_DPP_F32(v, 0x142, 0xA);   // row_bcast:15 into rows 1 and 3 (0 elsewhere)
  v += NEURST_DPP_F32(v, 0x143, 0xC);   // row_bcast:31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// block-wide sum for blockDim.x multiple of 64 (<= 1024); red must hold 16 floats
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
  return r;
}

// ---- 16x16 MFMA micro-kernel, templated on the operand type ----------------------
//  f32 : v_mfma_f32_16x16x4_f32   (exact fmaf chain; parity mode)
//  bf16: v_mfma_f32_16x16x32_bf16 (perf mode)
// Operand lane maps (cdna_hip_programming.md section 3):
//  bf16 A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15], j = 0..7
//  f32  A[row l&15][k = l>>4],      B[k = l>>4][col l&15]
//  C/D  col = l&15, row = 4(l>>4) + reg
template <typename T> struct Mma;
template <> struct Mma<float> {
  static constexpr int KS = 4;  // k per instruction
  typedef float Frag;
  // tile points at element [row0][k0] of a K-contiguous tile with leading dim ld (elements)
  static __device__ __forceinline__ Frag load(const float* tile, int ld, int lane) {
    return tile[(lane & 15) * ld + (lane >> 4)];
  }
  // element (row, k) lives at tile[k*ld + row]
  static __device__ __forceinline__ Frag load_kstrided(const float* tile, int ld, int lane) {
    return tile[(lane >> 4) * ld + (lane & 15)];
  }
  static __device__ __forceinline__ Frag load_tr(const float* tile, int ld, int lane) { return load_kstrided(tile, ld, lane); }
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<bf16_t> {
  static constexpr int KS = 32;
  typedef bf16x8 Frag;
  static __device__ __forceinline__ Frag load(const bf16_t* tile, int ld, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + (lane & 15) * ld + 8 * (lane >> 4));
  }
  static __device__ __forceinline__ Frag load_kstrided(const bf16_t* tile, int ld, int lane) {
    Frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = tile[(8 * (lane >> 4) + j) * ld + (lane & 15)];
    return f;
  }
  // Same fragment from a k-strided tile (element (row, k) at tile[k*ld + row], rows contiguous) with the
  // hardware transposing read ds_read_b64_tr_b16 (cdna_hip_programming.md T10): per 16-lane group g the
  // block is k rows 8g..8g+3 (then 8g+4..8g+7) x 16 consecutive rows-of-the-operand; lane 4q+p supplies the
  // address of block row q, columns 4p..4p+3, and lane i receives column i of the 4 block rows.
  // tile must be an LDS pointer, 8-byte aligned, ld*2 a multiple of 8; EXEC must be full.
  static __device__ __forceinline__ Frag load_tr(const bf16_t* tile, int ld, int lane) {
    typedef bf16x4 __attribute__((address_space(3))) * lds_p;
    const bf16_t* p = tile + (8 * (lane >> 4) + ((lane & 15) >> 2)) * ld + 4 * (lane & 3);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p + 4 * ld));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
