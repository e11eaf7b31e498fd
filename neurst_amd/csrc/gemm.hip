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
// the layout it has in memory: TRANS=false: memory [rows][K] -> LDS [128][LDK]; TRANS=true: memory [K][rows]
// -> LDS [BK][LDT].  Fragments of the second kind are read with the transposing LDS read.
template <typename T, bool TRANS>
struct Stager {
  static constexpr int BK = GemmCfg<T>::BK, EPC = GemmCfg<T>::EPC;
  static constexpr int LD = TRANS ? GemmCfg<T>::LDT : GemmCfg<T>::LDK;
  static constexpr int TILE = TRANS ? BK * GemmCfg<T>::LDT : 128 * GemmCfg<T>::LDK;   // elements
  static constexpr int NCH = 128 * BK / EPC / NT;
  static constexpr int CPR = TRANS ? 128 / EPC : BK / EPC;   // chunks per LDS row
  typename Chunk<T>::V reg[NCH];

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
      const int r = c / CPR, cc = (c % CPR) * EPC;
      *reinterpret_cast<typename Chunk<T>::V*>(S + r * LD + cc) = reg[i];
    }
  }
  // fragment of 16 output rows starting at `row` for the k-step starting at `ks`
  static __device__ __forceinline__ typename Mma<T>::Frag frag(const T* __restrict__ S, int row, int ks, int lane) {
    if constexpr (!TRANS) return Mma<T>::load(S + row * LD + ks, LD, lane);
    else return Mma<T>::load_tr(S + ks * LD + row, LD, lane);
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
  // TB_KCONTIG: B stored [N][K] (trans_b = 1); otherwise B stored [K][N].  TA: A stored [K][M].
  constexpr int BK = GemmCfg<T>::BK, KS = Mma<T>::KS;
  typedef Stager<T, TA> SA;
  typedef Stager<T, !TB_KCONTIG> SB;
  __shared__ __attribute__((aligned(16))) T As[2][SA::TILE];
  __shared__ __attribute__((aligned(16))) T Bs[2][SB::TILE];
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

  SA sa;
  SB sb;
  int cur = 0;
  if (kbeg < kend) {
    sa.load(A, g.lda, m0, kbeg, g.M, kend, g.a_aligned, tid);
    sb.load(B, g.ldb, n0, kbeg, g.N, kend, g.b_aligned, tid);
    sa.store(As[0], tid);
    sb.store(Bs[0], tid);
  }
  __syncthreads();
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {   // next K-tile: global -> registers, in flight while this tile is multiplied
      sa.load(A, g.lda, m0, k0 + BK, g.M, kend, g.a_aligned, tid);
      sb.load(B, g.ldb, n0, k0 + BK, g.N, kend, g.b_aligned, tid);
    }
    const T* Ac = As[cur];
    const T* Bc = Bs[cur];
#pragma unroll
    for (int ks = 0; ks < BK; ks += KS) {
      typename Mma<T>::Frag af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = SA::frag(Ac, wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = SB::frag(Bc, wn * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(af[i], bf[j], acc[i][j]);
    }
    if (more) {   // the other buffer was last read one iteration ago (barrier below): safe to refill
      sa.store(As[cur ^ 1], tid);
      sb.store(Bs[cur ^ 1], tid);
    }
    __syncthreads();
    cur ^= 1;
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
