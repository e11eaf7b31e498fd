// K3: C[M,N] = op(A) . op(B) (+bias)(relu)(relu-grad gate)(+C) on the matrix cores.
//   bf16 operands -> v_mfma_f32_16x16x32_bf16, fp32 accumulate   (perf mode)
//   fp32 operands -> v_mfma_f32_16x16x4_f32, exact fmaf chain     (1e-5 parity mode)
// Replaces `dense` (st/models/tools/utils.py:367-398), tf.layers.dense and their
// autodiff dgrad / wgrad.  128x128 block tile, 4 waves in 2x2, each wave 64x64 as 4x4
// MFMA tiles; LDS tiles are always K-contiguous (transposed operands are transposed
// while staging), register-prefetched one K-tile ahead.
#include "common.h"

template <typename T> struct GemmCfg;
template <> struct GemmCfg<float> {
  static constexpr int BK = 16, EPC = 4, LD = 20;
};
template <> struct GemmCfg<bf16_t> {
  static constexpr int BK = 64, EPC = 8, LD = 72;
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

// Stage one operand tile (128 rows x BK) into registers.  TRANS=false: memory is
// [rows][K] (K contiguous).  TRANS=true: memory is [K][rows].
template <typename T, bool TRANS>
struct Stager {
  static constexpr int BK = GemmCfg<T>::BK, EPC = GemmCfg<T>::EPC, LD = GemmCfg<T>::LD;
  static constexpr int NCH = 128 * BK / EPC / NT;
  typename Chunk<T>::V reg[NCH];

  __device__ __forceinline__ void load(const T* __restrict__ base, int64_t ld, int64_t row0, int64_t k0, int64_t R, int64_t Kend,
                                       bool aligned, int tid) {
    if (!TRANS) {
      constexpr int CPR = BK / EPC;  // chunks per row
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        int c = tid + i * NT;
        int row = c / CPR, kc = c % CPR;
        reg[i] = load_chunk<T>(base, ld, row0 + row, k0 + kc * EPC, R, Kend, aligned);
      }
    } else {
      constexpr int CPK = 128 / EPC;  // chunks per k-line
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        int kk, rc;
        map(i, tid, kk, rc);
        reg[i] = load_chunk<T>(base, ld, k0 + kk, row0 + rc * EPC, Kend, R, aligned);
      }
      (void)CPK;
    }
  }
  // transposed mapping: for bf16 chunks 2p and 2p+1 of a thread hold adjacent k lines
  static __device__ __forceinline__ void map(int i, int tid, int& kk, int& rc) {
    constexpr int CPK = 128 / EPC;
    if (sizeof(T) == 2) {
      int p = tid + (i >> 1) * NT;
      kk = 2 * (p / CPK) + (i & 1);
      rc = p % CPK;
    } else {
      int c = tid + i * NT;
      kk = c / CPK;
      rc = c % CPK;
    }
  }
  __device__ __forceinline__ void store(T* __restrict__ S, int tid) const {
    if constexpr (!TRANS) {
      constexpr int CPR = BK / EPC;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        int c = tid + i * NT;
        int row = c / CPR, kc = c % CPR;
        *reinterpret_cast<typename Chunk<T>::V*>(S + row * LD + kc * EPC) = reg[i];
      }
    } else if constexpr (sizeof(T) == 2) {
      typedef unsigned int __attribute__((may_alias)) u32_alias;
#pragma unroll
      for (int i = 0; i < NCH; i += 2) {
        int kk, rc;
        map(i, tid, kk, rc);
        const u32x4 a = __builtin_bit_cast(u32x4, reg[i]);       // k line kk   : 8 consecutive rows
        const u32x4 b = __builtin_bit_cast(u32x4, reg[i + 1]);   // k line kk+1 : same rows
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
          const unsigned int lo = (a[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
          const unsigned int hi = (b[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
          *reinterpret_cast<u32_alias*>(S + (rc * EPC + j) * LD + kk) = lo | (hi << 16);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        int kk, rc;
        map(i, tid, kk, rc);
#pragma unroll
        for (int j = 0; j < EPC; ++j) S[(rc * EPC + j) * LD + kk] = reg[i][j];
      }
    }
  }
};

struct GemmArgs {
  const void* A; const void* B; void* C; const float* bias; const void* relu_src;
  int64_t M, N, K, lda, ldb, ldc;
  int relu, accumulate, split_k, a_aligned, b_aligned;
  int tiles_m, tiles_n;
};

template <typename T, typename TC, bool TA, bool TB_KCONTIG>
__global__ void __launch_bounds__(NT) gemm_kernel(GemmArgs g) {
  // TB_KCONTIG: B stored [N][K] (trans_b = 1). Otherwise B stored [K][N] and is transposed while staging.
  constexpr int BK = GemmCfg<T>::BK, LD = GemmCfg<T>::LD, KS = Mma<T>::KS;
  __shared__ __attribute__((aligned(16))) T As[BM * LD];
  __shared__ __attribute__((aligned(16))) T Bs[BN * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware remap: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous run of tiles
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = bid % g.tiles_m, tn = bid / g.tiles_m;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  int64_t kbeg = 0, kend = g.K;
  if (g.split_k > 1) {
    int64_t per = ((g.K + g.split_k - 1) / g.split_k + BK - 1) / BK * BK;
    kbeg = per * blockIdx.z;
    kend = kbeg + per < g.K ? kbeg + per : g.K;
  }
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  Stager<T, TA> sa;
  Stager<T, !TB_KCONTIG> sb;
  if (kbeg < kend) {
    sa.load(A, g.lda, m0, kbeg, g.M, kend, g.a_aligned, tid);
    sb.load(B, g.ldb, n0, kbeg, g.N, kend, g.b_aligned, tid);
  }
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // previous tile fully consumed
    sa.store(As, tid);
    sb.store(Bs, tid);
    __syncthreads();
    if (k0 + BK < kend) {  // prefetch next K-tile into registers while this one is multiplied
      sa.load(A, g.lda, m0, k0 + BK, g.M, kend, g.a_aligned, tid);
      sb.load(B, g.ldb, n0, k0 + BK, g.N, kend, g.b_aligned, tid);
    }
#pragma unroll
    for (int ks = 0; ks < BK; ks += KS) {
      typename Mma<T>::Frag af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = Mma<T>::load(As + (wm * 64 + i * 16) * LD + ks, LD, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = Mma<T>::load(Bs + (wn * 64 + j * 16) * LD + ks, LD, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(af[i], bf[j], acc[i][j]);
    }
  }
  // epilogue
  TC* C = reinterpret_cast<TC*>(g.C);
  const T* rs = reinterpret_cast<const T*>(g.relu_src);
  const bool add_bias = g.bias && blockIdx.z == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t col = n0 + wn * 64 + j * 16 + (lane & 15);
      if (col >= g.N) continue;
      const float bv = add_bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (row >= g.M) continue;
        float v = acc[i][j][r] + bv;
        if (g.relu) v = fmaxf(v, 0.f);
        const int64_t off = row * g.ldc + col;
        if (rs) v = to_f32<T>(rs[off]) > 0.f ? v : 0.f;
        if (g.split_k > 1) {
          if (sizeof(TC) == 4) atomicAdd(reinterpret_cast<float*>(C) + off, v);
        } else {
          if (g.accumulate) v += to_f32<TC>(C[off]);
          C[off] = from_f32<TC>(v);
        }
      }
    }
  }
}

template <typename T, typename TC>
static int launch_gemm(const GemmArgs& g, int trans_a, int trans_b, hipStream_t s) {
  dim3 grid(g.tiles_m * g.tiles_n, 1, g.split_k);
  if (!trans_a && trans_b) gemm_kernel<T, TC, false, true><<<grid, NT, 0, s>>>(g);
  else if (!trans_a && !trans_b) gemm_kernel<T, TC, false, false><<<grid, NT, 0, s>>>(g);
  else if (trans_a && !trans_b) gemm_kernel<T, TC, true, false><<<grid, NT, 0, s>>>(g);
  else gemm_kernel<T, TC, true, true><<<grid, NT, 0, s>>>(g);
  return 0;
}

extern "C" int neurst_gemm_bias_act(const void* A, const void* B, void* C, const float* bias, const void* relu_src, int64_t M,
                                    int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int trans_a, int trans_b,
                                    int ab_dtype, int c_dtype, int relu, int accumulate, int split_k, void* stream) {
  NEURST_CHECK_ARG(A && B && C, "gemm: null operand");
  NEURST_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  NEURST_CHECK_ARG(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, "gemm: leading dimension too small");
  NEURST_CHECK_ARG(!(ab_dtype == NEURST_F32 && c_dtype == NEURST_BF16), "gemm: fp32 operands need an fp32 result");
  if (split_k < 1) split_k = 1;
  NEURST_CHECK_ARG(split_k == 1 || (c_dtype == NEURST_F32 && accumulate && !relu && !relu_src),
                   "gemm: split_k > 1 needs fp32 C, accumulate=1 and no activation");
  NEURST_CHECK_ARG(split_k <= 64, "gemm: split_k too large");
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.relu_src = relu_src;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.relu = relu; g.accumulate = accumulate; g.split_k = split_k;
  size_t es = dtype_size(ab_dtype);
  g.a_aligned = (((uintptr_t)A & 15) == 0) && ((lda * es) % 16 == 0);
  g.b_aligned = (((uintptr_t)B & 15) == 0) && ((ldb * es) % 16 == 0);
  int64_t tm = cdiv64(M, BM), tn = cdiv64(N, BN);
  NEURST_CHECK_ARG(tm * tn < (1ll << 30), "gemm: too many tiles");
  g.tiles_m = (int)tm; g.tiles_n = (int)tn;
  hipStream_t s = as_stream(stream);
  if (ab_dtype == NEURST_BF16) {
    if (c_dtype == NEURST_BF16) launch_gemm<bf16_t, bf16_t>(g, trans_a, trans_b, s);
    else launch_gemm<bf16_t, float>(g, trans_a, trans_b, s);
  } else {
    launch_gemm<float, float>(g, trans_a, trans_b, s);
  }
  NEURST_CHECK_LAUNCH("gemm_bias_act");
  return NEURST_OK;
}

// ---------------------------------------------------------------- column sums (bias gradients)
// grid (ceil(N/64), nsplit): each block sums a 64-column stripe over its share of the rows
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t M, int64_t N, int64_t ldx) {
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + lane;
  const int64_t rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = rows_per * blockIdx.y, r1 = r0 + rows_per < M ? r0 + rows_per : M;
  float s = 0.f;
  if (col < N)
    for (int64_t r = r0 + w; r < r1; r += 4) s += to_f32<T>(x[r * ldx + col]);
  sm[w][lane] = s;
  __syncthreads();
  if (w == 0 && col < N) atomicAdd(out + col, sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane]);
}
__global__ void zero_f32_kernel(float* p, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
extern "C" int neurst_colsum(const void* x, float* out, int64_t M, int64_t N, int64_t ldx, int dtype, int accumulate, void* stream) {
  NEURST_CHECK_ARG(M > 0 && N > 0 && ldx >= N, "colsum: bad shape");
  hipStream_t s = as_stream(stream);
  if (!accumulate) {
    zero_f32_kernel<<<(int)cdiv64(N, 256), 256, 0, s>>>(out, N);
    NEURST_CHECK_LAUNCH("colsum_zero");
  }
  int64_t nsplit = cdiv64(M, 256);
  if (nsplit > 64) nsplit = 64;
  dim3 grid((unsigned)cdiv64(N, 64), (unsigned)nsplit);
  if (dtype == NEURST_BF16) colsum_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, out, M, N, ldx);
  else colsum_kernel<float><<<grid, 256, 0, s>>>((const float*)x, out, M, N, ldx);
  NEURST_CHECK_LAUNCH("colsum");
  return NEURST_OK;
}

// ---------------------------------------------------------------- MFMA fragment self-test
template <typename T>
__global__ void selftest_mma_kernel(const T* __restrict__ A, const T* __restrict__ Bt, float* __restrict__ C, int K) {
  const int lane = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; k += Mma<T>::KS) acc = Mma<T>::mma(Mma<T>::load(A + k, K, lane), Mma<T>::load(Bt + k, K, lane), acc);
  for (int r = 0; r < 4; ++r) C[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
}
extern "C" int neurst_selftest_mma(const void* A, const void* Bt, float* C, int64_t K, int dtype, void* stream) {
  NEURST_CHECK_ARG(K > 0 && K % 32 == 0, "selftest_mma: K must be a positive multiple of 32");
  if (dtype == NEURST_BF16) selftest_mma_kernel<bf16_t><<<1, 64, 0, as_stream(stream)>>>((const bf16_t*)A, (const bf16_t*)Bt, C, (int)K);
  else selftest_mma_kernel<float><<<1, 64, 0, as_stream(stream)>>>((const float*)A, (const float*)Bt, C, (int)K);
  NEURST_CHECK_LAUNCH("selftest_mma");
  return NEURST_OK;
}
