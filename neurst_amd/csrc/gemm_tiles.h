// Pieces shared by the GEMM kernels of gemm.hip (128 x 128 tiles) and gemm256.hip (256 x 256 tiles, 8 waves):
// operand tile stagers (LDS layouts + fragment reads), the argument block and the LDS-staged epilogue.
#pragma once
#include "common.h"

template <typename T> struct GemmCfg;
template <> struct GemmCfg<float> {
  // BK x 4 B rows; LDK pads K-contiguous rows, LDT pads the 128-wide rows of k-strided tiles
  static constexpr int BK = 16, EPC = 4, LDK = 20, LDT = 144;
};
template <> struct GemmCfg<bf16_t> {
  static constexpr int BK = 64, EPC = 8, LDK = 72, LDT = 144;   // LDT: 288-B rows -> conflict-free ds_read_b64_tr_b16
};
constexpr int BM = 128, BN = 128, NT = 256;

template <typename T> struct Chunk;
template <> struct Chunk<float> { typedef f32x4 V; };
template <> struct Chunk<bf16_t> { typedef bf16x8 V; };

template <typename T>
__device__ __forceinline__ typename Chunk<T>::V load_chunk(const T* __restrict__ base, int64_t ld, int64_t r, int64_t c,
                                                           int64_t R, int64_t Cc, bool aligned) {
  constexpr int EPC = GemmCfg<T>::EPC;
  typename Chunk<T>::V v;
  if (r < R && c + EPC <= Cc && aligned) {
    v = *reinterpret_cast<const typename Chunk<T>::V*>(base + r * ld + c);
  } else {
#pragma unroll
    for (int j = 0; j < EPC; ++j) v[j] = (r < R && c + j < Cc) ? base[r * ld + c + j] : from_f32<T>(0.f);
  }
  return v;
}

// One operand tile (128 rows of the output dimension x BK of the reduction) staged global -> registers -> LDS in
// the layout it has in memory: TRANS=false: memory [rows][K] -> LDS [128][BK]; TRANS=true: memory [K][rows]
// -> LDS [BK][128].  Fragments of the second kind are read with the transposing LDS read.
// bf16 tiles are unpadded and XOR-swizzled at 16-byte granularity so that both ds_read_b128 row reads
// (128-B rows: chunk ^= row&7) and ds_read_b64_tr_b16 (256-B rows: chunk ^= ((row&3)<<2)|((row>>2)&3),
// cdna_hip_programming.md T10 image (b)) are bank-conflict free; fp32 (parity mode) tiles are padded.
template <typename T, bool TRANS, int BKT = GemmCfg<T>::BK, int ROWS = 128>
struct Stager {
  // ROWS = rows of the output dimension in the tile: 128, or 32 for the A side of the thin-tile variant (K-contiguous only)
  static_assert(ROWS == 128 || !TRANS, "k-strided tiles are 128 wide");
  static constexpr int BK = BKT, EPC = GemmCfg<T>::EPC;
  static constexpr bool SWZ = sizeof(T) == 2;
  static constexpr int LD = SWZ ? (TRANS ? 128 : BK) : (TRANS ? GemmCfg<T>::LDT : GemmCfg<T>::LDK);
  static constexpr int TILE = (TRANS ? BK : ROWS) * LD;   // elements
  static constexpr int NCH = ROWS * BK / EPC / NT;
  static_assert(NCH >= 1, "tile smaller than one 16-byte chunk per thread");
  static constexpr int CPR = TRANS ? 128 / EPC : BK / EPC;   // chunks per LDS row
  typename Chunk<T>::V reg[NCH];

  // element offset of 16-byte chunk `ch` of LDS row `r`
  // K-contiguous tiles: 128-B rows (BK = 64) spread a read group over the 8 chunk slots with r & 7; 64-B rows (BK = 32)
  // alias every 4 rows and use the 4 slots per row with (-(r >> 2)) & 3
  static __device__ __forceinline__ int kswz(int r) { return BK == 64 ? (r & 7) : ((0 - (r >> 2)) & 3); }
  static __device__ __forceinline__ int off(int r, int ch) {
    if constexpr (!SWZ) return r * LD + ch * EPC;
    else if constexpr (!TRANS) return r * LD + ((ch ^ kswz(r)) << 3);
    else return r * LD + ((ch ^ (((r & 3) << 2) | ((r >> 2) & 3))) << 3);
  }
  __device__ __forceinline__ void load(const T* __restrict__ base, int64_t ld, int64_t row0, int64_t k0, int64_t R, int64_t Kend,
                                       bool aligned, int tid) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * NT;
      const int r = c / CPR, cc = (c % CPR) * EPC;
      if constexpr (!TRANS) reg[i] = load_chunk<T>(base, ld, row0 + r, k0 + cc, R, Kend, aligned);
      else reg[i] = load_chunk<T>(base, ld, k0 + r, row0 + cc, Kend, R, aligned);
    }
  }
  __device__ __forceinline__ void store(T* __restrict__ S, int tid) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * NT;
      *reinterpret_cast<typename Chunk<T>::V*>(S + off(c / CPR, c % CPR)) = reg[i];
    }
  }
  // fragment of 16 output rows starting at `row` (multiple of 16) for the k-step starting at `ks`
  static __device__ __forceinline__ typename Mma<T>::Frag frag(const T* __restrict__ S, int row, int ks, int lane) {
    if constexpr (!SWZ) {
      if constexpr (!TRANS) return Mma<T>::load(S + row * LD + ks, LD, lane);
      else return Mma<T>::load_tr(S + ks * LD + row, LD, lane);
    } else if constexpr (!TRANS) {
      const int r = row + (lane & 15), ch = (ks >> 3) + (lane >> 4);
      return *reinterpret_cast<const bf16x8*>(S + off(r, ch));
    } else {
      typedef bf16x4 __attribute__((address_space(3))) * lds_p;
      const int r = ks + 8 * (lane >> 4) + ((lane & 15) >> 2);
      const int ch = (row >> 3) + ((lane >> 1) & 1), half = (lane & 1) << 2;
      bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(S + off(r, ch) + half));
      bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(S + off(r + 4, ch) + half));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }
};

struct GemmArgs {
  const void* A; const void* B; void* C; const float* bias; const void* relu_src;
  int64_t M, N, K, lda, ldb, ldc;
  int relu, accumulate, split_k, a_aligned, b_aligned, c_vec;
  int tiles_m, tiles_n;
  float* opsum;   // SUMOP 1: += sum_k op(A)[m][k] (length M); SUMOP 2: += sum_k op(B)[k][n] (length N)
};

template <typename T> __device__ __forceinline__ typename Mma<T>::Frag ones_frag();
template <> __device__ __forceinline__ float ones_frag<float>() { return 1.f; }
template <> __device__ __forceinline__ bf16x8 ones_frag<bf16_t>() {
  const bf16_t o = (bf16_t)1.f;
  return bf16x8{o, o, o, o, o, o, o, o};
}

// direct global -> LDS copy of one full operand tile (bf16, FAST path): every lane moves 16 B with
// global_load_lds_dwordx4; the LDS image of one wave-instruction is lane-linear, so the XOR swizzle is
// applied to the SOURCE chunk each lane fetches (cdna_hip_programming.md 5.4 rule 21).  Rows of the
// output dimension beyond R are clamped to a valid address: they only feed output rows/columns that the
// epilogue never stores.  The reduction range of the tile must be fully in bounds.
template <typename T, bool TRANS, int BKT, int ROWS = 128>
__device__ __forceinline__ void issue_glds(const T* __restrict__ base, int64_t ld, int64_t row0, int64_t k0, int64_t R,
                                           T* __restrict__ S, int tid) {
  typedef Stager<T, TRANS, BKT, ROWS> St;
#pragma unroll
  for (int i = 0; i < St::NCH; ++i) {
    const int P = tid + i * NT;                      // 16-byte chunk position in the LDS tile
    const int r = P / St::CPR, cpos = P % St::CPR;
    int c;                                           // source chunk that belongs at this position
    if constexpr (!TRANS) c = cpos ^ St::kswz(r);
    else c = cpos ^ (((r & 3) << 2) | ((r >> 2) & 3));
    const T* src;
    if constexpr (!TRANS) {
      int64_t gr = row0 + r;
      gr = gr < R ? gr : R - 1;
      src = base + gr * ld + k0 + c * 8;
    } else {
      int64_t gc = row0 + c * 8;
      gc = gc + 8 <= R ? gc : 0;
      src = base + (k0 + r) * ld + gc;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(S + (size_t)(i * NT + (tid & ~63)) * 8), 16, 0, 0);
  }
}

// plain stores: non-temporal C stores measured 1-5 % faster per launch in isolation but 0.6 % slower over the step
// (the consumer kernel then misses the memory-side cache)
#define STORE_C(ptr, val) (*(ptr) = (val))
constexpr int PEPI_LD = 36;                                   // fp32 row stride of a 16 x 32 staging half-tile
constexpr int PEPI_BYTES = 4 * 16 * PEPI_LD * 4;              // one per wave

// SIMPLE: the forward call of a dense layer and nothing else - no gate, no accumulation, whole 16-byte chunks (the host checks
// relu_src == NULL, accumulate == 0, c_vec and N % 8 == 0) - so the unrolled chunks are straight-line code with one row test each
template <typename T, typename TC, int MI = 4, bool SIMPLE = false>
__device__ __forceinline__ void persist_epilogue(const GemmArgs& g, f32x4 (&acc)[MI][4], const float (&bcol)[4], float* __restrict__ Cs,
                                                 int64_t wrow0, int64_t wcol0, int lane) {
  TC* C = reinterpret_cast<TC*>(g.C);
  const T* rs = SIMPLE ? nullptr : reinterpret_cast<const T*>(g.relu_src);
  const bool accumulate = SIMPLE ? false : g.accumulate != 0;
  const bool c_vec = SIMPLE ? true : g.c_vec != 0;
  const int lg = lane >> 4, l15 = lane & 15;
  const int rr = lane >> 2, cc = (lane & 3) * 8;
  // relu-gradient gate: all eight chunks this lane needs are requested up front (one wait instead of eight)
  typename Chunk<T>::V gate[2 * MI];
  if (rs) {
#pragma unroll
    for (int p = 0; p < 2 * MI; ++p) {
      const int64_t row = wrow0 + (p >> 1) * 16 + rr, col = wcol0 + (p & 1) * 32 + cc;
      if (row < g.M && c_vec && col + 8 <= g.N) gate[p] = *reinterpret_cast<const typename Chunk<T>::V*>(rs + row * g.ldc + col);
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
      // 16 rows x 32 columns of the wave's tile: MFMA layout -> LDS -> 8 consecutive columns per lane
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[i][jh * 2 + jj][r] + bcol[jh * 2 + jj];
          if (g.relu) v = fmaxf(v, 0.f);
          Cs[(lg * 4 + r) * PEPI_LD + jj * 16 + l15] = v;
        }
      // the same wave reads back what it wrote: LDS operations of one wave complete in order
      const f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + rr * PEPI_LD + cc);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + rr * PEPI_LD + cc + 4);
      const int64_t row = wrow0 + i * 16 + rr, col = wcol0 + jh * 32 + cc;
      if (row >= g.M || col >= g.N) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
      const int64_t off = row * g.ldc + col;
      if (SIMPLE || (c_vec && col + 8 <= g.N)) {
        if (rs) {
          const typename Chunk<T>::V m = gate[i * 2 + jh];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = to_f32<T>(m[e]) > 0.f ? v[e] : 0.f;
        }
        if constexpr (sizeof(TC) == 2) {
          if (accumulate) {
            const bf16x8 old = *reinterpret_cast<const bf16x8*>(C + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)old[e];
          }
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
          STORE_C(reinterpret_cast<bf16x8*>(C + off), o);
        } else {
          f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
          if (accumulate) {
            o0 += *reinterpret_cast<const f32x4*>(C + off);
            o1 += *reinterpret_cast<const f32x4*>(C + off + 4);
          }
          STORE_C(reinterpret_cast<f32x4*>(C + off), o0);
          STORE_C(reinterpret_cast<f32x4*>(C + off + 4), o1);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (col + e >= g.N) continue;
          float x = v[e];
          if (rs) x = to_f32<T>(rs[off + e]) > 0.f ? x : 0.f;
          if (accumulate) x += to_f32<TC>(C[off + e]);
          C[off + e] = from_f32<TC>(x);
        }
      }
    }
}


// gemm256.hip: the 256 x 256-tile persistent kernel for large grids (bf16 operands, split_k == 1)
bool neurst_gemm256_eligible(const GemmArgs& g, int trans_a, int trans_b, bool c_f32);
void neurst_gemm256_launch(const GemmArgs& g, int trans_a, int trans_b, int c_dtype, hipStream_t s);
bool neurst_gemm256_group_launch(int n, const GemmArgs* probs, const int* tile_begin, const int* tile_end, hipStream_t s);
bool neurst_gemm256_group_eligible(const GemmArgs& g);
int neurst_gemm256_group_max();
int64_t neurst_gemm256_group_tiles(int64_t M, int64_t N);
