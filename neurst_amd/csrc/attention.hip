// K4/K4c/K4x: fused multi-head attention forward + backward (flash style, the
// [B,H,Tq,Tk] matrix is never materialised).  Reference arithmetic:
// st/models/tools/utils.py:545-553 (q *= dh^-0.5, split/merge heads) and
// st/layers/common_attention_v1.py:180-204,325-365 (additive -1e9 biases, softmax,
// dropout on the weights, weights.v).
//
// Layout: one workgroup = 4 waves = 64 query rows (forward, dQ) or 64 keys (dK/dV);
// K/V (or Q/dO) tiles of 64 rows are staged through LDS in their natural [row][dh] layout; every
// product runs on 16x16 MFMA tiles (bf16: 16x16x32, fp32 parity mode: 16x16x4).  Scores are computed
// transposed so that each lane owns one query (or key): probabilities and dS stay in registers and feed
// the second product directly (no LDS round trip), row statistics need two shuffles per tile.
#include "attention2.h"

#define NEG_BIAS (-1e9f)
#ifndef DKV_PREFETCH
#define DKV_PREFETCH 0
#endif
#ifndef FWD_WPE
#define FWD_WPE 3
#endif
#ifndef DQ_WPE
#define DQ_WPE 3
#endif
#ifndef DKV_WPE
#define DKV_WPE 3
#endif
// head sizes above 128 (the KD attention, dh = 192) get two waves per SIMD: at three (168 registers) those instantiations
// spilled 80-130 registers per lane to scratch
#define ATT_OCC(n) __attribute__((amdgpu_waves_per_eu(n, n)))
// stops the instruction scheduler from hoisting every LDS fragment read of a phase to its start (which costs
// 100+ registers and the occupancy with them); other waves hide the LDS latency instead
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

template <typename T> struct ACfg;
template <> struct ACfg<float> { static constexpr int EPC = 4, PAD = 4; typedef f32x4 V; };
template <> struct ACfg<bf16_t> { static constexpr int EPC = 8, PAD = 16; typedef bf16x8 V; };   // (DH + 16) / 2 dwords = 8 mod 16: conflict-free b128 and tr reads

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
// The same tile in two halves - global -> registers, registers -> LDS - so that the loads of the NEXT tile are in
// flight while the current one is multiplied.  FULL (dh == DH, 16-byte aligned rows): branch-free 16-byte loads, rows
// beyond the tensor are clamped and zeroed by a select.
template <typename T, int DH, bool FULL>
struct TileRegs {
  static constexpr int EPC = ACfg<T>::EPC, LD = DH + ACfg<T>::PAD, CPR = DH / EPC, NCH = 64 * CPR / 256;
  typename ACfg<T>::V reg[NCH];
  __device__ __forceinline__ void load(const T* __restrict__ g, int64_t ldg, int64_t row0, int64_t nrows, int dh, bool aligned, int tid) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * 256, r = c / CPR, cc = (c % CPR) * EPC;
      if constexpr (FULL) {
        const int64_t row = row0 + r;
        const typename ACfg<T>::V v = *reinterpret_cast<const typename ACfg<T>::V*>(g + (row < nrows ? row : nrows - 1) * ldg + cc);
        const typename ACfg<T>::V zero = {};
        reg[i] = row < nrows ? v : zero;
      } else {
        reg[i] = load_row_chunk<T>(g + (row0 + r) * ldg, cc, dh, row0 + r < nrows, aligned);
      }
    }
  }
  __device__ __forceinline__ void store(T* __restrict__ S, int tid) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * 256;
      *reinterpret_cast<typename ACfg<T>::V*>(S + (c / CPR) * LD + (c % CPR) * EPC) = reg[i];
    }
  }
};

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

// ---- register-resident probabilities -------------------------------------------------------------------
// Scores are produced TRANSPOSED so that the probabilities never visit LDS.  A 16x16 accumulator tile holds,
// in lane l, column (l & 15) and rows 4(l >> 4) + r.  With the contraction index of the SECOND product on the
// accumulator rows, the four registers of nt-tile pairs are exactly a B-operand fragment of that product once
// its k slots are permuted: slot (g = l >> 4, j) of k-step `step` stands for tile row
//     bf16 (k = 32 per step): 32 step + 16 (j >> 2) + 4 g + (j & 3)      fp32 (k = 4): 16 (step >> 2) + 4 g + (step & 3)
// and the other operand (rows of a natural [row][col] LDS tile, read transposed) uses the same permutation.
template <typename T> struct RegP;
template <> struct RegP<float> {
  static constexpr int NSTEP = 16;
  static __device__ __forceinline__ float bfrag(const f32x4 (&p)[4], int step) { return p[step >> 2][step & 3]; }
  // same fragment from the two nt tiles of one half (steps NSTEP/2 * half + ls)
  static __device__ __forceinline__ float bfrag_half(const f32x4 (&p)[2], int ls) { return p[ls >> 2][ls & 3]; }
  // A fragment: element (row = col0 + (l & 15), slot) = tile[tile_row(slot)][col0 + (l & 15)]
  static __device__ __forceinline__ float afrag_tr(const float* tile, int ld, int step, int col0, int lane) {
    return tile[(16 * (step >> 2) + 4 * (lane >> 4) + (step & 3)) * ld + col0 + (lane & 15)];
  }
};
template <> struct RegP<bf16_t> {
  static constexpr int NSTEP = 2;
  static __device__ __forceinline__ bf16x8 bfrag(const f32x4 (&p)[4], int step) {
    const f32x4 a = p[2 * step], b = p[2 * step + 1];
    return bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
  }
  static __device__ __forceinline__ bf16x8 bfrag_half(const f32x4 (&p)[2], int) {
    const f32x4 a = p[0], b = p[1];
    return bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
  }
  // two ds_read_b64_tr_b16: within a 16-lane group, lane 4q+p addresses tile row (base + q), columns 4p..4p+3 and
  // lane i receives column i of those 4 rows (EXEC must be full; 8-byte aligned)
  static __device__ __forceinline__ bf16x8 afrag_tr(const bf16_t* tile, int ld, int step, int col0, int lane) {
    typedef bf16x4 __attribute__((address_space(3))) * lds_p;
    const bf16_t* p = tile + (32 * step + 4 * (lane >> 4) + ((lane & 15) >> 2)) * ld + col0 + 4 * (lane & 3);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p + 16 * ld));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
};

// softmax arithmetic: fp32 = exact expf in natural units (parity mode); bf16 = base-2 with log2(e) folded into the scale
template <typename T> struct SM;
template <> struct SM<float> {
  static __device__ __forceinline__ float unit() { return 1.f; }
  static __device__ __forceinline__ float ex(float x) { return expf(x); }
  static __device__ __forceinline__ float lse(float m, float l) { return m + logf(l); }
};
template <> struct SM<bf16_t> {
  static __device__ __forceinline__ float unit() { return 1.4426950408889634f; }
  static __device__ __forceinline__ float ex(float x) { return __builtin_amdgcn_exp2f(x); }
  static __device__ __forceinline__ float lse(float m, float l) { return (m + __log2f(l)) * 0.6931471805599453f; }
};

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
  uint32_t thr, key, inner0;   // thr: 16-bit keep threshold (0 = no dropout); inner0 = fmix32(key): the inner hash when every counter fits 32 bits
  int idx32;
  float keep_scale, qscale;
  int q_aligned, k_aligned, v_aligned, o_aligned;
  int out_vec, dq_vec, dk_vec, dv_vec;   // 4 consecutive head columns may be stored as one vector
  // optional [B*H][Tq][ceil(Tk/32)] u32, row-major bitmap of the dropout keep decisions: bit (k & 31) of word k >> 5 of row
  // (b*H + h)*Tq + q.  Written by the forward, read back by the backward kernels instead of re-hashing every index.
  uint32_t* keep_bits;
  int nw;                // words per bitmap row
  int keep_bits_ready;   // forward: keep_bits already holds the decisions (neurst_mha_keep_bits): load them instead of hashing
  int delta_local;       // backward, one launch for dQ and dK / dV: the dK / dV workgroups compute delta themselves
  int nq_blocks, nq_tiles, nk_tiles;   // that launch: its first nq_blocks blocks are dQ workgroups; query / key tiles per (batch, head)
};

// XCD-aware block placement: workgroups are dealt round-robin to the 8 XCDs in dispatch order, so the tiles that share one
// (batch, head)'s K/V (or Q/dO) would land on 8 different L2s and fetch them 8 times over the fabric.  Give each XCD a
// contiguous run of the flattened (tile, head, batch) index instead (bijective for any grid size).
struct BlockId { int64_t x, y, z; };
__device__ __forceinline__ BlockId remap_block() {
  const int nx = gridDim.x, ny = gridDim.y, total = nx * ny * (int)gridDim.z;
  const int lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
  const int q = total >> 3, r = total & 7, xcd = lin & 7, loc = lin >> 3;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  return BlockId{id % nx, (id / nx) % ny, id / (nx * ny)};
}

// The two keep words (keys kv0 .. kv0 + 63) of attention row `row` (= (b*H + h)*Tq + q), generated here: lanes l, l^16, l^32,
// l^48 share a row - the even 16-lane groups draw the first word, the odd ones the second, one shuffle exchanges them.
__device__ __forceinline__ void gen_tile_words(const AttnArgs& a, int lane, int64_t row, int64_t kv0, uint32_t& w0, uint32_t& w1) {
  const int odd = (lane >> 4) & 1;
  uint32_t mine = attn_keep_word(a.key, a.inner0, a.idx32, (uint64_t)(row * a.nw + (kv0 >> 5) + odd), a.thr);
  asm volatile("" : "+v"(mine));                              // keeps the generator out of the MFMA phases' schedule
  const uint32_t other = __shfl_xor(mine, 16, 64);
  w0 = odd ? other : mine;
  w1 = odd ? mine : other;
}
// The 16 decisions a lane of 16-lane group g owns in a 64-key tile (bit nt*4 + r <-> key kv0 + 16 nt + 4 g + r) <-> the two bitmap
// words of the tile: word (nt >> 1), bits 16 (nt & 1) + 4 g + r.
__device__ __forceinline__ void kmask_to_words(uint32_t kmask, int g, uint32_t& w0, uint32_t& w1) {
  w0 = ((kmask & 0xFu) << (4 * g)) | (((kmask >> 4) & 0xFu) << (16 + 4 * g));
  w1 = (((kmask >> 8) & 0xFu) << (4 * g)) | (((kmask >> 12) & 0xFu) << (16 + 4 * g));
}
__device__ __forceinline__ uint32_t words_to_kmask(uint32_t w0, uint32_t w1, int g) {
  return ((w0 >> (4 * g)) & 0xFu) | (((w0 >> (16 + 4 * g)) & 0xFu) << 4) | (((w1 >> (4 * g)) & 0xFu) << 8) | (((w1 >> (16 + 4 * g)) & 0xFu) << 12);
}
// every lane of a row holds both words: the g == 0 lane stores them
__device__ __forceinline__ void store_tile_words(const AttnArgs& a, uint32_t w0, uint32_t w1, int lane, int64_t row, bool row_ok, int64_t kv0) {
  if (lane < 16 && row_ok) {
    uint32_t* p = a.keep_bits + row * a.nw + (kv0 >> 5);
    p[0] = w0;
    if ((kv0 >> 5) + 1 < a.nw) p[1] = w1;
  }
}
// this lane's 16 decisions of the tile, generated (and stored when there is a bitmap to fill) or all-ones without dropout
__device__ __forceinline__ uint32_t gen_kmask(const AttnArgs& a, int lane, int64_t row, bool row_ok, int64_t kv0, bool store) {
  if (!a.thr) return 0xFFFFu;
  uint32_t w0, w1;
  gen_tile_words(a, lane, row, kv0, w0, w1);
  if (store) store_tile_words(a, w0, w1, lane, row, row_ok, kv0);
  return words_to_kmask(w0, w1, lane >> 4);
}
__device__ __forceinline__ uint32_t load_keep_words(const AttnArgs& a, int lane, int64_t row, bool row_ok, int64_t kv0) {
  if (!row_ok) return 0u;
  const uint32_t* p = a.keep_bits + row * a.nw + (kv0 >> 5);
  const uint32_t w0 = p[0], w1 = (kv0 >> 5) + 1 < a.nw ? p[1] : 0u;
  return words_to_kmask(w0, w1, lane >> 4);
}

__device__ __forceinline__ float quad_groups_max(float v) { return fmaxf(fmaxf(v, __shfl_xor(v, 16, 64)), fmaxf(__shfl_xor(v, 32, 64), __shfl_xor(v, 48, 64))); }
__device__ __forceinline__ float quad_groups_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// scaled score of (query qi, key kj) with the reference's additive biases (in units of `unit`); -inf outside the tensor
__device__ __forceinline__ float biased_score(float raw, float c, float unit, int64_t qi, int64_t kj, int64_t Tk, int mask_mode, int len) {
  if (kj >= Tk) return -INFINITY;
  float s = raw * c;
  if (mask_mode == NEURST_MASK_KEYPAD) { if (kj >= len) s += NEG_BIAS * unit; }
  else if (mask_mode == NEURST_MASK_CAUSAL) { if (kj > qi) s += NEG_BIAS * unit; }
  return s;
}

// 4 consecutive head columns c0..c0+3 of one row
template <typename T>
__device__ __forceinline__ void store4(T* __restrict__ row, int c0, int dh, f32x4 v, bool vec) {
  if (vec && c0 + 4 <= dh) {
    if constexpr (sizeof(T) == 4) *reinterpret_cast<f32x4*>(row + c0) = v;
    else *reinterpret_cast<bf16x4*>(row + c0) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) if (c0 + r < dh) row[c0 + r] = from_f32<T>(v[r]);
  }
}

// ================================================================== forward
// workgroup = 64 queries (16 per wave); lane l works for query (l & 15); S^T = K.Q^T, O^T = V^T.P^T
template <typename T, int DH, bool FULL>
__global__ void __launch_bounds__(256) ATT_OCC((DH > 128 ? 2 : FWD_WPE)) mha_fwd_kernel(AttnArgs a) {
  constexpr int KS = Mma<T>::KS, LDK = DH + ACfg<T>::PAD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Ks = reinterpret_cast<T*>(smem);            // [64][LDK]
  T* Vs = Ks + 64 * LDK;                         // [64][LDK]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g4 = (lane >> 4) * 4;
  const BlockId bid = remap_block();
  const int64_t b = bid.z, h = bid.y, q0 = bid.x * 64;
  const int dh = a.dh;
  const T* qg = reinterpret_cast<const T*>(a.q) + b * a.bsq + h * dh;
  const T* kg = reinterpret_cast<const T*>(a.k) + b * a.bsk + h * dh;
  const T* vg = reinterpret_cast<const T*>(a.v) + b * a.bsv + h * dh;
  const int len = a.mask_mode == NEURST_MASK_KEYPAD ? a.len_k[b] : 0;
  int64_t kv_end = a.Tk;
  if (a.mask_mode == NEURST_MASK_CAUSAL) kv_end = q0 + 64 < a.Tk ? q0 + 64 : a.Tk;
  else if (a.mask_mode == NEURST_MASK_KEYPAD && len > 0 && len < a.Tk) kv_end = len;
  const int64_t kv_lim = a.mask_mode == NEURST_MASK_KEYPAD ? (len < a.Tk ? len : a.Tk) : a.Tk;
  const float unit = SM<T>::unit(), c = a.qscale * unit;

  typename Mma<T>::Frag qf[DH / KS];
  load_frags_global<T, DH>(qf, qg, a.ldq, q0 + w * 16, a.Tq, dh, a.q_aligned, lane);

  f32x4 oacc[DH / 16];   // O^T: lane = query (l & 15), rows = head columns 16 dt + g4 + r
#pragma unroll
  for (int i = 0; i < DH / 16; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const int64_t bh = b * a.H + h, qi = q0 + w * 16 + (lane & 15);

  TileRegs<T, DH, FULL> kreg, vreg;
  kreg.load(kg, a.ldk, 0, a.Tk, dh, a.k_aligned, tid);
  vreg.load(vg, a.ldv, 0, a.Tk, dh, a.v_aligned, tid);
  for (int64_t kv0 = 0; kv0 < kv_end; kv0 += 64) {
    __syncthreads();
    kreg.store(Ks, tid);
    vreg.store(Vs, tid);
    __syncthreads();
    if (kv0 + 64 < kv_end) {   // next tile: in flight while this one is multiplied
      kreg.load(kg, a.ldk, kv0 + 64, a.Tk, dh, a.k_aligned, tid);
      vreg.load(vg, a.ldv, kv0 + 64, a.Tk, dh, a.v_aligned, tid);
    }
    f32x4 s[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      s[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DH / KS; ++ks)
        s[nt] = Mma<T>::mma(Mma<T>::load(Ks + nt * 16 * LDK + ks * KS, LDK, lane), qf[ks], s[nt]);
      SCHED_FENCE();
    }
    const bool edge = a.mask_mode == NEURST_MASK_CAUSAL || kv0 + 64 > kv_lim;   // wave-uniform: tile touches a bias or the tensor end
    float mloc = -INFINITY;
    if (edge) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[nt][r] = biased_score(s[nt][r], c, unit, qi, kv0 + nt * 16 + g4 + r, a.Tk, a.mask_mode, len);
          mloc = fmaxf(mloc, s[nt][r]);
        }
    } else {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[nt][r] *= c; mloc = fmaxf(mloc, s[nt][r]); }
    }
    const float mn = fmaxf(m_run, quad_groups_max(mloc));
    const float alpha = (m_run == -INFINITY) ? 0.f : SM<T>::ex(m_run - mn);
    m_run = mn;
    float lsum = 0.f;
    uint32_t kmask;
    if (a.keep_bits_ready) {
      kmask = load_keep_words(a, lane, bh * a.Tq + qi, qi < a.Tq, kv0);
    } else {
      kmask = gen_kmask(a, lane, bh * a.Tq + qi, qi < a.Tq, kv0, a.keep_bits != nullptr);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = (edge && s[nt][r] == -INFINITY) ? 0.f : SM<T>::ex(s[nt][r] - mn);
        lsum += p;
        if (a.thr) p = ((kmask >> (nt * 4 + r)) & 1u) ? p * a.keep_scale : 0.f;
        s[nt][r] = p;
      }
    l_run = l_run * alpha + lsum;
#pragma unroll
    for (int i = 0; i < DH / 16; ++i) oacc[i] *= alpha;
#pragma unroll
    for (int st = 0; st < RegP<T>::NSTEP; ++st) {
      const typename Mma<T>::Frag pf = RegP<T>::bfrag(s, st);
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        oacc[dt] = Mma<T>::mma(RegP<T>::afrag_tr(Vs, LDK, st, dt * 16, lane), pf, oacc[dt]);
        if (dt & 1) SCHED_FENCE();
      }
    }
  }
  const float l = quad_groups_sum(l_run);
  const float inv = 1.f / l;
  if (qi < a.Tq) {
    T* orow = reinterpret_cast<T*>(a.out) + b * a.bso + h * dh + qi * a.ldo;
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) store4<T>(orow, dt * 16 + g4, dh, oacc[dt] * inv, a.out_vec);
    if (lane < 16) a.lse[bh * a.Tq + qi] = SM<T>::lse(m_run, l) ;
  }
}

// ================================================================== backward: delta + dQ
// lane l works for query (l & 15): S^T = K.Q^T, dP^T = V.dO^T, dQ^T = K^T.dS^T; also writes delta = rowsum(dO * O)
template <typename T, int DH, bool FULL>
__device__ __forceinline__ void mha_bwd_dq_body(const AttnArgs& a, const BlockId bid, char* smem) {
  constexpr int KS = Mma<T>::KS, LDK = DH + ACfg<T>::PAD;
  T* Ks = reinterpret_cast<T*>(smem);             // [64][LDK]
  T* Vs = Ks + 64 * LDK;                          // [64][LDK]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g4 = (lane >> 4) * 4;
  const int64_t b = bid.z, h = bid.y, q0 = bid.x * 64;
  const int dh = a.dh;
  const T* qg = reinterpret_cast<const T*>(a.q) + b * a.bsq + h * dh;
  const T* kg = reinterpret_cast<const T*>(a.k) + b * a.bsk + h * dh;
  const T* vg = reinterpret_cast<const T*>(a.v) + b * a.bsv + h * dh;
  const T* og = reinterpret_cast<const T*>(a.o) + b * a.bso + h * dh;
  const T* dog = reinterpret_cast<const T*>(a.d_o) + b * a.bso + h * dh;
  const int len = a.mask_mode == NEURST_MASK_KEYPAD ? a.len_k[b] : 0;
  int64_t kv_end = a.Tk;
  if (a.mask_mode == NEURST_MASK_CAUSAL) kv_end = q0 + 64 < a.Tk ? q0 + 64 : a.Tk;
  else if (a.mask_mode == NEURST_MASK_KEYPAD && len > 0 && len < a.Tk) kv_end = len;
  const int64_t kv_lim = a.mask_mode == NEURST_MASK_KEYPAD ? (len < a.Tk ? len : a.Tk) : a.Tk;
  const int64_t bh = b * a.H + h, qi = q0 + w * 16 + (lane & 15);
  const bool q_tail = q0 + 64 > a.Tq;
  const float unit = SM<T>::unit(), c = a.qscale * unit;

  typename Mma<T>::Frag qf[DH / KS], dof[DH / KS];
  load_frags_global<T, DH>(qf, qg, a.ldq, q0 + w * 16, a.Tq, dh, a.q_aligned, lane);
  load_frags_global<T, DH>(dof, dog, a.ldo, q0 + w * 16, a.Tq, dh, a.o_aligned, lane);
  float del;
  {
    typename Mma<T>::Frag of[DH / KS];
    load_frags_global<T, DH>(of, og, a.ldo, q0 + w * 16, a.Tq, dh, a.o_aligned, lane);
    float d = 0.f;
#pragma unroll
    for (int ks = 0; ks < DH / KS; ++ks) {
      if constexpr (sizeof(T) == 4) d += of[ks] * dof[ks];
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) d += (float)of[ks][j] * (float)dof[ks][j];
      }
    }
    del = quad_groups_sum(d);
    if (lane < 16 && qi < a.Tq) a.delta[bh * a.Tq + qi] = del;
  }
  const float lse_q = (qi < a.Tq ? a.lse_in[bh * a.Tq + qi] : 0.f) * unit;
  f32x4 dqacc[DH / 16];
#pragma unroll
  for (int i = 0; i < DH / 16; ++i) dqacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  TileRegs<T, DH, FULL> kreg, vreg;
  kreg.load(kg, a.ldk, 0, a.Tk, dh, a.k_aligned, tid);
  vreg.load(vg, a.ldv, 0, a.Tk, dh, a.v_aligned, tid);
  for (int64_t kv0 = 0; kv0 < kv_end; kv0 += 64) {
    __syncthreads();
    kreg.store(Ks, tid);
    vreg.store(Vs, tid);
    __syncthreads();
    if (kv0 + 64 < kv_end) {
      kreg.load(kg, a.ldk, kv0 + 64, a.Tk, dh, a.k_aligned, tid);
      vreg.load(vg, a.ldv, kv0 + 64, a.Tk, dh, a.v_aligned, tid);
    }
    const bool edge = a.mask_mode == NEURST_MASK_CAUSAL || kv0 + 64 > kv_lim || q_tail;
    const uint32_t kmask = (a.keep_bits && a.thr) ? load_keep_words(a, lane, bh * a.Tq + qi, qi < a.Tq, kv0)
                                                  : gen_kmask(a, lane, bh * a.Tq + qi, qi < a.Tq, kv0, false);
    f32x4 ds[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DH / KS; ++ks) {
        s = Mma<T>::mma(Mma<T>::load(Ks + nt * 16 * LDK + ks * KS, LDK, lane), qf[ks], s);
        dp = Mma<T>::mma(Mma<T>::load(Vs + nt * 16 * LDK + ks * KS, LDK, lane), dof[ks], dp);
      }
      SCHED_FENCE();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p;
        if (edge) {
          const float sc = biased_score(s[r], c, unit, qi, kv0 + nt * 16 + g4 + r, a.Tk, a.mask_mode, len);
          p = (sc == -INFINITY || qi >= a.Tq) ? 0.f : SM<T>::ex(sc - lse_q);
        } else {
          p = SM<T>::ex(s[r] * c - lse_q);
        }
        float dpe = dp[r];
        if (a.thr) dpe = ((kmask >> (nt * 4 + r)) & 1u) ? dpe * a.keep_scale : 0.f;
        ds[nt][r] = p * (dpe - del);
      }
    }
#pragma unroll
    for (int st = 0; st < RegP<T>::NSTEP; ++st) {
      const typename Mma<T>::Frag dsf = RegP<T>::bfrag(ds, st);
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        dqacc[dt] = Mma<T>::mma(RegP<T>::afrag_tr(Ks, LDK, st, dt * 16, lane), dsf, dqacc[dt]);
        if (dt & 1) SCHED_FENCE();
      }
    }
  }
  if (qi < a.Tq) {
    T* dqrow = reinterpret_cast<T*>(a.dq) + b * a.Tq * a.lddq + h * dh + qi * a.lddq;
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) store4<T>(dqrow, dt * 16 + g4, dh, dqacc[dt] * a.qscale, a.dq_vec);
  }
}
template <typename T, int DH, bool FULL>
__global__ void __launch_bounds__(256) ATT_OCC((DH > 128 ? 2 : DQ_WPE)) mha_bwd_dq_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  mha_bwd_dq_body<T, DH, FULL>(a, remap_block(), smem);
}

// ================================================================== backward: dK, dV
// workgroup = 64 keys (16 per wave); lane l works for key (l & 15): S = Q.K^T, dP = dO.V^T, dV^T = dO^T.P, dK^T = Q^T.dS
template <typename T, int DH, bool FULL>
__device__ __forceinline__ void mha_bwd_dkv_body(const AttnArgs& a, const BlockId bid, char* smem) {
  constexpr int KS = Mma<T>::KS, LDK = DH + ACfg<T>::PAD;
  T* Qs = reinterpret_cast<T*>(smem);                 // [64][LDK]
  T* dOs = Qs + 64 * LDK;                             // [64][LDK]
  float* lse_s = reinterpret_cast<float*>(dOs + 64 * LDK);  // [64]
  float* del_s = lse_s + 64;                          // [64]
  uint32_t* mask_s = reinterpret_cast<uint32_t*>(del_s + 64);   // [2][64]: the two bitmap words of this 64-key tile for each of the 64 staged queries
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g4 = (lane >> 4) * 4;
  const int64_t b = bid.z, h = bid.y, kv0 = bid.x * 64;
  const int dh = a.dh;
  const T* qg = reinterpret_cast<const T*>(a.q) + b * a.bsq + h * dh;
  const T* kg = reinterpret_cast<const T*>(a.k) + b * a.bsk + h * dh;
  const T* vg = reinterpret_cast<const T*>(a.v) + b * a.bsv + h * dh;
  const T* dog = reinterpret_cast<const T*>(a.d_o) + b * a.bso + h * dh;
  const int len = a.mask_mode == NEURST_MASK_KEYPAD ? a.len_k[b] : 0;
  const int64_t bh = b * a.H + h, kj = kv0 + w * 16 + (lane & 15);
  const bool dead = a.mask_mode == NEURST_MASK_KEYPAD && len > 0 && kv0 >= len;  // padded keys get exactly zero weight
  const float unit = SM<T>::unit(), c = a.qscale * unit;

  typename Mma<T>::Frag kf[DH / KS], vf[DH / KS];
  load_frags_global<T, DH>(kf, kg, a.ldk, kv0 + w * 16, a.Tk, dh, a.k_aligned, lane);
  load_frags_global<T, DH>(vf, vg, a.ldv, kv0 + w * 16, a.Tk, dh, a.v_aligned, lane);
  f32x4 dkacc[DH / 16], dvacc[DH / 16];
#pragma unroll
  for (int i = 0; i < DH / 16; ++i) { dkacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  int64_t q_begin = 0;
  if (a.mask_mode == NEURST_MASK_CAUSAL) q_begin = kv0;  // queries before the first key of this tile are fully masked
  if (dead) q_begin = a.Tq;
  const int64_t kv_lim = a.mask_mode == NEURST_MASK_KEYPAD ? (len < a.Tk ? len : a.Tk) : a.Tk;

  TileRegs<T, DH, FULL> qreg, doreg;
  float lse_reg = 0.f, del_reg = 0.f;   // threads 0..63: statistics of query q0 + tid of the staged tile
  uint32_t mask_reg = 0;                // threads 64..191: bitmap word (tid - 64) >> 6 of query (tid - 64) & 63
  const bool stored = a.thr != 0;       // dropout: the words come from the bitmap, or are drawn again here when there is none
  auto load_stats = [&](int64_t q0n) {
    const int64_t qi = q0n + tid;
    if (tid < 64 && qi < a.Tq) {
      lse_reg = a.lse_in[bh * a.Tq + qi] * unit;
      if (a.delta_local) {      // fused launch: the dQ blocks that write delta run beside this one - rowsum(dO * O) again here
        const T* orow = reinterpret_cast<const T*>(a.o) + b * a.bso + h * dh + qi * a.ldo;
        const T* drow = dog + qi * a.ldo;
        float d = 0.f;
        if (sizeof(T) == 2 && a.o_aligned) {
          for (int e = 0; e < dh; e += 8) {
            const bf16x8 o8 = *reinterpret_cast<const bf16x8*>(orow + e), d8 = *reinterpret_cast<const bf16x8*>(drow + e);
#pragma unroll
            for (int j = 0; j < 8; ++j) d += (float)o8[j] * (float)d8[j];
          }
        } else {
          for (int e = 0; e < dh; ++e) d += to_f32<T>(orow[e]) * to_f32<T>(drow[e]);
        }
        del_reg = d;
      } else {
        del_reg = a.delta[bh * a.Tq + qi];
      }
    } else { lse_reg = 0.f; del_reg = 0.f; }
    if (stored && tid >= 64 && tid < 192) {
      const int64_t qm = q0n + ((tid - 64) & 63), wi = (kv0 >> 5) + ((tid - 64) >> 6);
      if (qm < a.Tq && wi < a.nw)
        mask_reg = a.keep_bits ? a.keep_bits[(bh * a.Tq + qm) * a.nw + wi]
                               : attn_keep_word(a.key, a.inner0, a.idx32, (uint64_t)((bh * a.Tq + qm) * a.nw + wi), a.thr);
      else mask_reg = 0u;
    }
  };
  if (q_begin < a.Tq) {
    qreg.load(qg, a.ldq, q_begin, a.Tq, dh, a.q_aligned, tid);
    doreg.load(dog, a.ldo, q_begin, a.Tq, dh, a.o_aligned, tid);
    load_stats(q_begin);
  }
  for (int64_t q0 = q_begin; q0 < a.Tq; q0 += 64) {
#if !DKV_PREFETCH
    if (q0 > q_begin) {
      qreg.load(qg, a.ldq, q0, a.Tq, dh, a.q_aligned, tid);
      doreg.load(dog, a.ldo, q0, a.Tq, dh, a.o_aligned, tid);
      load_stats(q0);
    }
#endif
    __syncthreads();
    qreg.store(Qs, tid);
    doreg.store(dOs, tid);
    if (tid < 64) { lse_s[tid] = lse_reg; del_s[tid] = del_reg; }
    else if (stored && tid < 192) mask_s[tid - 64] = mask_reg;
    __syncthreads();
#if DKV_PREFETCH
    if (q0 + 64 < a.Tq) {
      qreg.load(qg, a.ldq, q0 + 64, a.Tq, dh, a.q_aligned, tid);
      doreg.load(dog, a.ldo, q0 + 64, a.Tq, dh, a.o_aligned, tid);
      load_stats(q0 + 64);
    }
#endif
    const bool edge = a.mask_mode == NEURST_MASK_CAUSAL || kv0 + 64 > kv_lim || q0 + 64 > a.Tq;
    // weight (q = q0 + 16 nt + g4 + r, key kj): hashed here, or bit (kj & 31) of the query's bitmap word - the words of four
    // neighbouring queries come with one 16-byte LDS read per nt
    uint32_t kmask;
    if (stored) {
      kmask = 0;
      const int sh = (int)(kj & 31);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const u32x4 m4 = *reinterpret_cast<const u32x4*>(mask_s + (w >> 1) * 64 + nt * 16 + g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) kmask |= ((m4[r] >> sh) & 1u) << (nt * 4 + r);
      }
    } else {
      kmask = 0xFFFFu;
    }
    // two halves of 32 queries: the probabilities of a half are consumed by the second products right away
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      f32x4 pacc[2], dsacc[2];
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2) {
        const int nt = half * 2 + n2;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < DH / KS; ++ks) {
          s = Mma<T>::mma(Mma<T>::load(Qs + nt * 16 * LDK + ks * KS, LDK, lane), kf[ks], s);
          dp = Mma<T>::mma(Mma<T>::load(dOs + nt * 16 * LDK + ks * KS, LDK, lane), vf[ks], dp);
        }
        SCHED_FENCE();
        const int qloc = nt * 16 + g4;
        const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + qloc), del4 = *reinterpret_cast<const f32x4*>(del_s + qloc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t qi = q0 + qloc + r;
          float p;
          if (edge) {
            const float sc = biased_score(s[r], c, unit, qi, kj, a.Tk, a.mask_mode, len);
            p = (sc == -INFINITY || qi >= a.Tq) ? 0.f : SM<T>::ex(sc - lse4[r]);
          } else {
            p = SM<T>::ex(s[r] * c - lse4[r]);
          }
          float pd = p, dpe = dp[r];
          if (a.thr) {
            const bool keep = (kmask >> (nt * 4 + r)) & 1u;
            pd = keep ? p * a.keep_scale : 0.f;
            dpe = keep ? dpe * a.keep_scale : 0.f;
          }
          pacc[n2][r] = pd;
          dsacc[n2][r] = p * (dpe - del4[r]);
        }
      }
#pragma unroll
      for (int ls = 0; ls < RegP<T>::NSTEP / 2; ++ls) {
        const int st = half * (RegP<T>::NSTEP / 2) + ls;
        const typename Mma<T>::Frag pf = RegP<T>::bfrag_half(pacc, ls), dsf = RegP<T>::bfrag_half(dsacc, ls);
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) {
          dvacc[dt] = Mma<T>::mma(RegP<T>::afrag_tr(dOs, LDK, st, dt * 16, lane), pf, dvacc[dt]);
          dkacc[dt] = Mma<T>::mma(RegP<T>::afrag_tr(Qs, LDK, st, dt * 16, lane), dsf, dkacc[dt]);
          SCHED_FENCE();
        }
      }
    }
  }
  if (kj < a.Tk) {
    T* dkrow = reinterpret_cast<T*>(a.dk) + b * a.Tk * a.lddk + h * dh + kj * a.lddk;
    T* dvrow = reinterpret_cast<T*>(a.dv) + b * a.Tk * a.lddv + h * dh + kj * a.lddv;
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) {
      store4<T>(dkrow, dt * 16 + g4, dh, dkacc[dt] * a.qscale, a.dk_vec);
      store4<T>(dvrow, dt * 16 + g4, dh, dvacc[dt], a.dv_vec);
    }
  }
}
template <typename T, int DH, bool FULL>
__global__ void __launch_bounds__(256) ATT_OCC((DH > 128 ? 2 : DKV_WPE)) mha_bwd_dkv_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  mha_bwd_dkv_body<T, DH, FULL>(a, remap_block(), smem);
}
// Short-query shapes (Tq <= 256: decoder self / cross attention, KD attention): dQ and dK / dV in ONE launch.  Both are chains of
// dependent tiles on a few hundred workgroups that leave the chip mostly idle; side by side they cost the longer of the two
// instead of the sum.  Blocks 0 .. nq_blocks - 1 are the dQ workgroups (64 queries of one (head, batch) each), the rest the dK / dV workgroups
// (64 keys each), which compute delta = rowsum(dO * O) of every query tile they stage themselves (a.delta_local).
template <typename T, int DH, bool FULL>
__global__ void __launch_bounds__(256) ATT_OCC(2) mha_bwd_small_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lin = blockIdx.x;
  if (lin < a.nq_blocks) {
    const int nqt = a.nq_tiles;
    mha_bwd_dq_body<T, DH, FULL>(a, BlockId{lin % nqt, (lin / nqt) % (int)a.H, lin / (nqt * (int)a.H)}, smem);
  } else {
    const int id = lin - a.nq_blocks, nx = a.nk_tiles;
    mha_bwd_dkv_body<T, DH, FULL>(a, BlockId{id % nx, (id / nx) % (int)a.H, id / (nx * (int)a.H)}, smem);
  }
}

// ================================================================== host side
// neurst_set_option("mha_small_fused", 0) / NEURST_MHA_SMALL_FUSED=0: dQ and dK / dV of short-query shapes as two launches
int g_opt_mha_small_fused = [] { const char* e = getenv("NEURST_MHA_SMALL_FUSED"); return e ? atoi(e) : 1; }();
template <typename T, int DH> static size_t fwd_smem() { return sizeof(T) * 2 * 64 * (DH + ACfg<T>::PAD); }
template <typename T, int DH> static size_t dq_smem() { return sizeof(T) * 2 * 64 * (DH + ACfg<T>::PAD); }
template <typename T, int DH> static size_t dkv_smem() { return sizeof(T) * 2 * 64 * (DH + ACfg<T>::PAD) + 128 * sizeof(float) + 128 * sizeof(uint32_t); }

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
  a.thr = rate > 0.f ? attn_threshold16(rate) : 0u;   // keep iff the 16-bit draw is below it (common.h:attn_keep_word); 0 = no dropout
  a.key = key; a.keep_scale = keep_scale(rate);
  a.nw = (int)cdiv64(Tk, 32);
  a.inner0 = fmix32(key);   // == fmix32(hi ^ key) with hi == 0
  a.idx32 = (double)B * (double)H * (double)Tq * (double)a.nw * 4.0 < 4294967296.0;   // counters 4 widx + j of every keep word
  a.qscale = 1.0f / sqrtf((float)dh);
  size_t es = dtype_size(dtype);
  auto al = [&](const void* p, int64_t ld) { return (int)((((uintptr_t)p & 15) == 0) && ((ld * es) % 16 == 0) && ((dh * es) % 16 == 0)); };
  a.q_aligned = al(a.q, a.ldq); a.k_aligned = al(a.k, a.ldk); a.v_aligned = al(a.v, a.ldv);
  a.o_aligned = a.d_o ? (al(a.d_o, a.ldo) && al(a.o, a.ldo)) : 1;
  auto v4 = [&](const void* p, int64_t ld, int64_t bs) { return (int)(p && (((uintptr_t)p % (4 * es)) == 0) && (ld % 4 == 0) && (bs % 4 == 0) && (dh % 4 == 0)); };
  a.out_vec = v4(a.out, a.ldo, a.bso); a.dq_vec = v4(a.dq, a.lddq, 0); a.dk_vec = v4(a.dk, a.lddk, 0); a.dv_vec = v4(a.dv, a.lddv, 0);
  return NEURST_OK;
}

#define DISPATCH_ONE(T, KERNEL, SMEMFN, GRID, NAME, D)                                                            \
  rc = (full && dh == D) ? launch_attn(KERNEL<T, D, true>, GRID, SMEMFN<T, D>(), a, s, NAME)                      \
                         : launch_attn(KERNEL<T, D, false>, GRID, SMEMFN<T, D>(), a, s, NAME)
#define DISPATCH_DH(T, KERNEL, SMEMFN, GRID, NAME)                                                       \
  do {                                                                                                   \
    const bool full = sizeof(T) == 2 && a.q_aligned && a.k_aligned && a.v_aligned && a.o_aligned;        \
    if (dh <= 32) DISPATCH_ONE(T, KERNEL, SMEMFN, GRID, NAME, 32);                                       \
    else if (dh <= 64) DISPATCH_ONE(T, KERNEL, SMEMFN, GRID, NAME, 64);                                  \
    else if (dh <= 96) DISPATCH_ONE(T, KERNEL, SMEMFN, GRID, NAME, 96);                                  \
    else if (dh <= 128) DISPATCH_ONE(T, KERNEL, SMEMFN, GRID, NAME, 128);                                \
    else DISPATCH_ONE(T, KERNEL, SMEMFN, GRID, NAME, 192);                                               \
  } while (0)

// The forward's dropout decisions ahead of time: same grid, same block placement and same lane -> weight assignment as
// mha_fwd_kernel, nothing but the hash.  Depends on the key and the shape only, so it can run on another stream while
// the projection GEMM in front of the attention still works (the vector pipes idle under a GEMM).
__global__ void __launch_bounds__(256) mha_keep_bits_kernel(AttnArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g4 = (lane >> 4) * 4;
  const BlockId bid = remap_block();
  const int64_t bh = bid.z * a.H + bid.y, qi = bid.x * 64 + w * 16 + (lane & 15);
  for (int64_t kv0 = 0; kv0 < a.Tk; kv0 += 64)
    gen_kmask(a, lane, bh * a.Tq + qi, qi < a.Tq, kv0, true);
}

extern "C" int neurst_mha_keep_bits(uint16_t* keep_bits, int64_t B, int64_t H, int64_t Tq, int64_t Tk, float rate, uint32_t key, void* stream) {
  NEURST_CHECK_ARG(keep_bits && ((uintptr_t)keep_bits & 3) == 0, "mha_keep_bits: keep_bits must be a 4-byte aligned buffer");
  NEURST_CHECK_ARG(rate > 0.f && rate < 1.f, "mha_keep_bits: dropout rate must be in (0, 1)");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.keep_bits = reinterpret_cast<uint32_t*>(keep_bits);
  a.ldq = a.ldk = a.ldv = a.ldo = 1;
  int rc = fill_args(a, B, H, Tq, Tk, 1, NEURST_MASK_NONE, nullptr, rate, key, NEURST_BF16);
  if (rc) return rc;
  dim3 grid((unsigned)cdiv64(Tq, 64), (unsigned)H, (unsigned)B);
  mha_keep_bits_kernel<<<grid, 256, 0, as_stream(stream)>>>(a);
  NEURST_CHECK_LAUNCH("mha_keep_bits");
  return NEURST_OK;
}

extern "C" int64_t neurst_mha_keep_bits_bytes(int64_t B, int64_t H, int64_t Tq, int64_t Tk) {
  return B * H * Tq * cdiv64(Tk, 32) * (int64_t)sizeof(uint32_t);
}

static int mha_fwd_impl(const void* q, const void* k, const void* v, void* o, float* lse, uint16_t* keep_bits, int keep_bits_ready, int64_t ldq, int64_t ldk,
                        int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso, int64_t B, int64_t H,
                        int64_t Tq, int64_t Tk, int64_t dh, int mask_mode, const int32_t* len_k, float rate, uint32_t key,
                        int dtype, void* stream) {
  NEURST_CHECK_ARG(q && k && v && o && lse, "mha_fwd: null pointer");
  NEURST_CHECK_ARG(((uintptr_t)keep_bits & 3) == 0, "mha_fwd: keep_bits must be 4-byte aligned");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.keep_bits = reinterpret_cast<uint32_t*>(keep_bits);
  a.keep_bits_ready = keep_bits_ready;
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
  {
    Mha2FwdProblem p = {q, k, v, o, lse, a.keep_bits, keep_bits_ready, ldq, ldk, ldv, ldo, a.bsq, a.bsk, a.bsv, a.bso, B, H, Tq, Tk, dh,
                        mask_mode, len_k, rate, key, dtype};
    if (neurst_mha2_fwd_eligible(p)) return neurst_mha2_fwd_launch(p, s);
  }
  dim3 grid((unsigned)cdiv64(Tq, 64), (unsigned)H, (unsigned)B);
  if (dtype == NEURST_BF16) DISPATCH_DH(bf16_t, mha_fwd_kernel, fwd_smem, grid, "mha_fwd");
  else DISPATCH_DH(float, mha_fwd_kernel, fwd_smem, grid, "mha_fwd");
  return rc;
}
extern "C" int neurst_mha_fwd_strided(const void* q, const void* k, const void* v, void* o, float* lse, int64_t ldq, int64_t ldk,
                                      int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso, int64_t B, int64_t H,
                                      int64_t Tq, int64_t Tk, int64_t dh, int mask_mode, const int32_t* len_k, float rate, uint32_t key,
                                      int dtype, void* stream) {
  return mha_fwd_impl(q, k, v, o, lse, nullptr, 0, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, B, H, Tq, Tk, dh, mask_mode, len_k, rate, key, dtype, stream);
}
extern "C" int neurst_mha_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int64_t ldq, int64_t ldk, int64_t ldv,
                              int64_t ldo, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh, int mask_mode,
                              const int32_t* len_k, float rate, uint32_t key, int dtype, void* stream) {
  return mha_fwd_impl(q, k, v, o, lse, nullptr, 0, ldq, ldk, ldv, ldo, Tq * ldq, Tk * ldk, Tk * ldv, Tq * ldo, B, H, Tq, Tk, dh, mask_mode,
                      len_k, rate, key, dtype, stream);
}
extern "C" int neurst_mha_fwd_train(const void* q, const void* k, const void* v, void* o, float* lse, uint16_t* keep_bits, int64_t ldq,
                                    int64_t ldk, int64_t ldv, int64_t ldo, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh,
                                    int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype, void* stream) {
  return mha_fwd_impl(q, k, v, o, lse, keep_bits, 0, ldq, ldk, ldv, ldo, Tq * ldq, Tk * ldk, Tk * ldv, Tq * ldo, B, H, Tq, Tk, dh, mask_mode,
                      len_k, rate, key, dtype, stream);
}
extern "C" int neurst_mha_fwd_pregen(const void* q, const void* k, const void* v, void* o, float* lse, const uint16_t* keep_bits, int64_t ldq,
                                     int64_t ldk, int64_t ldv, int64_t ldo, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh,
                                     int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype, void* stream) {
  NEURST_CHECK_ARG(keep_bits && rate > 0.f, "mha_fwd_pregen: needs the decisions of neurst_mha_keep_bits and a dropout rate > 0");
  return mha_fwd_impl(q, k, v, o, lse, const_cast<uint16_t*>(keep_bits), 1, ldq, ldk, ldv, ldo, Tq * ldq, Tk * ldk, Tk * ldv, Tq * ldo, B, H, Tq,
                      Tk, dh, mask_mode, len_k, rate, key, dtype, stream);
}

extern "C" int neurst_mha_bwd_train(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                                    const uint16_t* keep_bits, float* delta, void* dq, void* dk, void* dv, int64_t ldq, int64_t ldk,
                                    int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, int64_t B, int64_t H, int64_t Tq,
                                    int64_t Tk, int64_t dh, int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype,
                                    void* stream);
extern "C" int neurst_mha_bwd_train_ws(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                                       const uint16_t* keep_bits, float* delta, void* dq, void* dk, void* dv, int64_t ldq, int64_t ldk,
                                       int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, int64_t B, int64_t H, int64_t Tq,
                                       int64_t Tk, int64_t dh, int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype,
                                       void* workspace, int64_t workspace_bytes, void* stream);
extern "C" int neurst_mha_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                              float* delta, void* dq, void* dk, void* dv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                              int64_t lddq, int64_t lddk, int64_t lddv, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh,
                              int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype, void* stream) {
  return neurst_mha_bwd_train(q, k, v, o, d_o, lse, nullptr, delta, dq, dk, dv, ldq, ldk, ldv, ldo, lddq, lddk, lddv, B, H, Tq, Tk, dh,
                              mask_mode, len_k, rate, key, dtype, stream);
}
extern "C" int neurst_mha_bwd_train(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                                    const uint16_t* keep_bits, float* delta, void* dq, void* dk, void* dv, int64_t ldq, int64_t ldk,
                                    int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, int64_t B, int64_t H, int64_t Tq,
                                    int64_t Tk, int64_t dh, int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype,
                                    void* stream) {
  return neurst_mha_bwd_train_ws(q, k, v, o, d_o, lse, keep_bits, delta, dq, dk, dv, ldq, ldk, ldv, ldo, lddq, lddk, lddv, B, H, Tq, Tk, dh,
                                 mask_mode, len_k, rate, key, dtype, nullptr, 0, stream);
}
extern "C" int neurst_mha_bwd_train_ws(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                                       const uint16_t* keep_bits, float* delta, void* dq, void* dk, void* dv, int64_t ldq, int64_t ldk,
                                       int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, int64_t B, int64_t H, int64_t Tq,
                                       int64_t Tk, int64_t dh, int mask_mode, const int32_t* len_k, float rate, uint32_t key, int dtype,
                                       void* workspace, int64_t workspace_bytes, void* stream) {
  NEURST_CHECK_ARG(q && k && v && o && d_o && lse && delta && dq && dk && dv, "mha_bwd: null pointer");
  NEURST_CHECK_ARG(((uintptr_t)keep_bits & 3) == 0, "mha_bwd: keep_bits must be 4-byte aligned");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.keep_bits = reinterpret_cast<uint32_t*>(const_cast<uint16_t*>(keep_bits));
  a.q = q; a.k = k; a.v = v; a.o = o; a.d_o = d_o; a.lse_in = lse; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  int rc = fill_args(a, B, H, Tq, Tk, dh, mask_mode, len_k, rate, key, dtype);
  if (rc) return rc;
  NEURST_CHECK_ARG(ldq >= H * dh && ldk >= H * dh && ldv >= H * dh && ldo >= H * dh && lddq >= H * dh && lddk >= H * dh && lddv >= H * dh,
                   "mha_bwd: row stride smaller than H*dh");
  hipStream_t s = as_stream(stream);
  {
    Mha2Problem p = {q, k, v, o, d_o, lse, a.keep_bits, delta, dq, dk, dv, ldq, ldk, ldv, ldo, lddq, lddk, lddv, B, H, Tq, Tk, dh,
                     mask_mode, len_k, rate, dtype};
    NEURST_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "mha_bwd: workspace must be 16-byte aligned");
    if (neurst_mha2_bwd_eligible(p, workspace ? workspace_bytes : 0)) return neurst_mha2_bwd_launch(p, static_cast<float*>(workspace), s);
  }
  dim3 gq((unsigned)cdiv64(Tq, 64), (unsigned)H, (unsigned)B);
  dim3 gk((unsigned)cdiv64(Tk, 64), (unsigned)H, (unsigned)B);
  // (two waves per SIMD: every instantiation spills less than its two stand-alone kernels do at three)
  if (Tq <= 256 && g_opt_mha_small_fused && B * H * (cdiv64(Tq, 64) + cdiv64(Tk, 64)) < (1ll << 30)) {
    a.delta_local = 1;
    a.nq_tiles = (int)cdiv64(Tq, 64);
    a.nq_blocks = (int)(B * H * a.nq_tiles);
    a.nk_tiles = (int)cdiv64(Tk, 64);
    dim3 gs((unsigned)(B * H * (a.nq_tiles + a.nk_tiles)));
    if (dtype == NEURST_BF16) DISPATCH_DH(bf16_t, mha_bwd_small_kernel, dkv_smem, gs, "mha_bwd_small");
    else DISPATCH_DH(float, mha_bwd_small_kernel, dkv_smem, gs, "mha_bwd_small");
    return rc;
  }
  if (dtype == NEURST_BF16) DISPATCH_DH(bf16_t, mha_bwd_dq_kernel, dq_smem, gq, "mha_bwd_dq");
  else DISPATCH_DH(float, mha_bwd_dq_kernel, dq_smem, gq, "mha_bwd_dq");
  if (rc) return rc;
  if (dtype == NEURST_BF16) DISPATCH_DH(bf16_t, mha_bwd_dkv_kernel, dkv_smem, gk, "mha_bwd_dkv");
  else DISPATCH_DH(float, mha_bwd_dkv_kernel, dkv_smem, gk, "mha_bwd_dkv");
  return rc;
}
