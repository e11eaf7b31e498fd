// K3: C[M,N] = op(A) . op(B) (+bias)(relu)(relu-grad gate)(+C) on the matrix cores.
//   bf16 operands -> v_mfma_f32_16x16x32_bf16, fp32 accumulate   (perf mode)
//   fp32 operands -> v_mfma_f32_16x16x4_f32, exact fmaf chain     (1e-5 parity mode)
// Replaces `dense` (st/models/tools/utils.py:367-398), tf.layers.dense and their
// autodiff dgrad / wgrad.  128x128 block tile, 4 waves in 2x2, each wave 64x64 as 4x4
// MFMA tiles.  Operand tiles sit in LDS in their memory layout (k-strided tiles are read with the
// transposing LDS read); aligned bf16 problems stream them global -> LDS directly, double-buffered.
// The weight-gradient entry point also produces the bias gradient (a sum of one operand over the
// reduction axis) with one extra MFMA against a fragment of ones per duty K-tile.
#include "common.h"
#include "gemm_tiles.h"

template <typename T, typename SA, typename SB, int SUMOP, int MI = 4>
__device__ __forceinline__ void mma_tile(const T* __restrict__ Ac, const T* __restrict__ Bc, f32x4 (&acc)[MI][4], f32x4 (&sacc)[4], bool duty,
                                         int wm, int wn, int lane) {
  // MI = 16-row MFMA tiles per wave along M: 4 (64 x 64 wave tile) or 1 (16 x 64, thin-tile variant)
  static_assert(MI == 4 || SUMOP == 0, "operand sums ride on the 64 x 64 wave tile");
  constexpr int BK = SA::BK, KS = Mma<T>::KS;
  typename Mma<T>::Frag sel = ones_frag<T>();
  if (!duty) sel = sel - sel;
#pragma unroll
  for (int ks = 0; ks < BK; ks += KS) {
    typename Mma<T>::Frag af[MI], bf[4];
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = SA::frag(Ac, wm * (16 * MI) + i * 16, ks, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[j] = SB::frag(Bc, wn * 64 + j * 16, ks, lane);
    if constexpr (sizeof(T) == 2) __builtin_amdgcn_s_setprio(1);   // a wave with its fragments in registers issues its MFMAs ahead of its neighbours' loads
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(af[i], bf[j], acc[i][j]);
    if constexpr (sizeof(T) == 2) __builtin_amdgcn_s_setprio(0);
    // the sum rides on the fragments already in registers: one more MFMA per fragment against ones (duty) or
    // zeros (not duty) - branch-free, so the register allocation of the main loop is unchanged
    if constexpr (SUMOP == 1) {
#pragma unroll
      for (int i = 0; i < MI; ++i) sacc[i] = Mma<T>::mma(af[i], sel, sacc[i]);
    } else if constexpr (SUMOP == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) sacc[j] = Mma<T>::mma(sel, bf[j], sacc[j]);
    }
  }
}

constexpr int EPI_LD = 68;                                   // fp32 row stride of the per-wave epilogue tile
constexpr int EPI_BYTES = 4 * 32 * EPI_LD * 4;               // 4 waves x 32 rows x 68 floats

// waves per SIMD the register allocation must leave room for: 3 for the bf16 NN kernel (K-tiles of 32: 32 KiB of LDS per
// workgroup, three of them hide each other's load latency), 1 elsewhere (LDS already limits those to one or two)
template <typename T, bool TA, bool TB_KCONTIG, bool FAST, int NS>
constexpr int gemm_min_waves() { return (sizeof(T) == 2 && !TA && !TB_KCONTIG && FAST && NS == 2) ? 3 : 1; }
// EPI: what the interior store passes know at compile time.  0: gate and accumulation are run-time flags; 1: neither (the forward
// call of a dense layer); 2: C += only; 3: ReLU gate only (the host picks by relu_src / accumulate).  Every run-time flag in the
// unrolled passes costs the K = 768 problems of the step 2-5 % (measured: see DESIGN.md, round 3).
template <typename T, typename TC, bool TA, bool TB_KCONTIG, bool FAST, int SUMOP = 0, int NS = 2, int EPI = 0>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(gemm_min_waves<T, TA, TB_KCONTIG, FAST, NS>())))
gemm_kernel(GemmArgs g) {
  // TB_KCONTIG: B stored [N][K] (trans_b = 1); otherwise B stored [K][N].  TA: A stored [K][M].
  // FAST (bf16, aligned operands): full K-tiles go global -> LDS directly (no VGPR staging).
  // NS = LDS stages.  2: three workgroups per CU hide each other's load latency (large grids).  4: for grids of at most
  // one workgroup per CU - three K-tiles (96 KB) stay in flight per CU across raw barriers with counted vmcnt waits.
  // K-tile depth: 64 (bf16), except NN (A K-contiguous, B k-strided) where 32 measured 3-14 % faster on the LUT shapes
  // (NT, TN and the 4-stage small-grid variant lose 5-25 % at 32); 16 for fp32
  constexpr int BK = (sizeof(T) == 2 && !TA && !TB_KCONTIG && NS == 2) ? 32 : GemmCfg<T>::BK;
  typedef Stager<T, TA, BK> SA;
  typedef Stager<T, !TB_KCONTIG, BK> SB;
  constexpr int STAGE_BYTES = NS * (SA::TILE + SB::TILE) * (int)sizeof(T);
  constexpr int SMEM = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];   // ONE LDS object (staging ring, then epilogue tiles)
  T* const lds = reinterpret_cast<T*>(smem);
  auto As = [&](int b) -> T* { return lds + b * SA::TILE; };
  auto Bs = [&](int b) -> T* { return lds + NS * SA::TILE + b * SB::TILE; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware remap: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous run of tiles
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  // grouped order: 8 tile-rows x all tile-columns at a time, rows fastest, so the ~64 workgroups an XCD runs
  // together share 8 A panels and 8 B panels in its 4 MiB L2 instead of streaming 64 A panels through it
  constexpr int GM = 8;
  const int per_group = GM * g.tiles_n;
  const int group = bid / per_group, first_m = group * GM;
  const int gsz = g.tiles_m - first_m < GM ? g.tiles_m - first_m : GM;
  const int tm = first_m + (bid % per_group) % gsz, tn = (bid % per_group) / gsz;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  int64_t kbeg = 0, kend = g.K;
  if (g.split_k > 1) {
    int64_t per = ((g.K + g.split_k - 1) / g.split_k + BK - 1) / BK * BK;
    kbeg = per * blockIdx.z;
    kend = kbeg + per < g.K ? kbeg + per : g.K;
    if (kbeg > kend) kbeg = kend;
  }
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // operand sum: the K-tiles are dealt round-robin to the blocks that share this operand panel, so every
  // tile is summed exactly once and no block carries more than its share of the extra MFMAs
  f32x4 sacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) sacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto duty = [&](int64_t k0) -> bool {
    if constexpr (SUMOP == 1) return wn == 0 && (int)((k0 / BK) % g.tiles_n) == tn;
    else if constexpr (SUMOP == 2) return wm == 0 && (int)((k0 / BK) % g.tiles_m) == tm;
    else return false;
  };

  int cur = 0;
  if constexpr (FAST) {
    const int nfull = (int)((kend - kbeg) / BK);
    if constexpr (NS == 2) {
      if (nfull > 0) {
        issue_glds<T, TA, BK>(A, g.lda, m0, kbeg, g.M, As(0), tid);
        issue_glds<T, !TB_KCONTIG, BK>(B, g.ldb, n0, kbeg, g.N, Bs(0), tid);
      }
      for (int it = 0; it < nfull; ++it) {
        __syncthreads();   // tile `it` has landed (vmcnt(0) precedes the barrier) and buf[cur^1] is no longer read
        if (it + 1 < nfull) {
          const int64_t kn = kbeg + (int64_t)(it + 1) * BK;
          issue_glds<T, TA, BK>(A, g.lda, m0, kn, g.M, As(cur ^ 1), tid);
          issue_glds<T, !TB_KCONTIG, BK>(B, g.ldb, n0, kn, g.N, Bs(cur ^ 1), tid);
        }
        mma_tile<T, SA, SB, SUMOP>(As(cur), Bs(cur), acc, sacc, duty(kbeg + (int64_t)it * BK), wm, wn, lane);
        cur ^= 1;
      }
    } else {
      // every thread issues LPT direct-to-LDS loads per K-tile; vmcnt counts them in order, so "at most LPT * n
      // outstanding" means everything but the newest n tiles has landed
      constexpr int LPT = SA::NCH + SB::NCH;
      static_assert(LPT * (NS - 2) <= 63, "vmcnt field");
      const int npro = nfull < NS - 1 ? nfull : NS - 1;
      for (int t = 0; t < npro; ++t) {
        issue_glds<T, TA, BK>(A, g.lda, m0, kbeg + (int64_t)t * BK, g.M, As(t), tid);
        issue_glds<T, !TB_KCONTIG, BK>(B, g.ldb, n0, kbeg + (int64_t)t * BK, g.N, Bs(t), tid);
      }
      for (int it = 0; it < nfull; ++it) {
        const int issued = it + NS - 1 < nfull ? it + NS - 1 : nfull;
        const int inflight = issued - it - 1;          // tiles newer than `it` already issued: 0 .. NS-2
        if (inflight >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT * 2) : "memory");
        else if (inflight == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // every wave's share of tile `it` is in LDS; buffer (it-1)%NS is free
        __builtin_amdgcn_sched_barrier(0);
        if (it + NS - 1 < nfull) {
          const int nb = (cur + NS - 1) % NS;
          const int64_t kn = kbeg + (int64_t)(it + NS - 1) * BK;
          issue_glds<T, TA, BK>(A, g.lda, m0, kn, g.M, As(nb), tid);
          issue_glds<T, !TB_KCONTIG, BK>(B, g.ldb, n0, kn, g.N, Bs(nb), tid);
        }
        mma_tile<T, SA, SB, SUMOP>(As(cur), Bs(cur), acc, sacc, duty(kbeg + (int64_t)it * BK), wm, wn, lane);
        cur = (cur + 1) % NS;
      }
    }
    const int64_t ktail = kbeg + (int64_t)nfull * BK;
    if (ktail < kend) {   // partial K-tile: register staging with zero fill
      SA sa;
      SB sb;
      sa.load(A, g.lda, m0, ktail, g.M, kend, true, tid);
      sb.load(B, g.ldb, n0, ktail, g.N, kend, true, tid);
      __syncthreads();
      sa.store(As(cur), tid);
      sb.store(Bs(cur), tid);
      __syncthreads();
      mma_tile<T, SA, SB, SUMOP>(As(cur), Bs(cur), acc, sacc, duty(ktail), wm, wn, lane);
    }
  } else {
    SA sa;
    SB sb;
    if (kbeg < kend) {
      sa.load(A, g.lda, m0, kbeg, g.M, kend, g.a_aligned, tid);
      sb.load(B, g.ldb, n0, kbeg, g.N, kend, g.b_aligned, tid);
      sa.store(As(0), tid);
      sb.store(Bs(0), tid);
    }
    __syncthreads();
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
      const bool more = k0 + BK < kend;
      if (more) {   // next K-tile: global -> registers, in flight while this tile is multiplied
        sa.load(A, g.lda, m0, k0 + BK, g.M, kend, g.a_aligned, tid);
        sb.load(B, g.ldb, n0, k0 + BK, g.N, kend, g.b_aligned, tid);
      }
      mma_tile<T, SA, SB, SUMOP>(As(cur), Bs(cur), acc, sacc, duty(k0), wm, wn, lane);
      if (more) {   // the other buffer was last read one iteration ago (barrier below): safe to refill
        sa.store(As(cur ^ 1), tid);
        sb.store(Bs(cur ^ 1), tid);
      }
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- epilogue: accumulators -> per-wave fp32 LDS tile (32 rows at a time) -> row-contiguous global stores
  __syncthreads();   // every wave is done with the staging buffers
  float* Cs = reinterpret_cast<float*>(smem) + wave * 32 * EPI_LD;
  TC* C = reinterpret_cast<TC*>(g.C);
  const T* rs = reinterpret_cast<const T*>(g.relu_src);
  const bool add_bias = g.bias && blockIdx.z == 0;
  const int64_t wrow0 = m0 + wm * 64, wcol0 = n0 + wn * 64;
  // the columns a lane stores are the same in every pass: fetch its bias values once, ahead of the store loops
  // (64 dependent scalar loads inside the loops used to cost more than the main loop of a K = 768 problem)
  float bv[8], bsplit = 0.f;
  {
    const int64_t col = wcol0 + (lane & 7) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = 0.f;
    if (add_bias) {
      if (g.split_k > 1) {
        if (wcol0 + lane < g.N) bsplit = g.bias[wcol0 + lane];
      } else if (col + 8 <= g.N && (((uintptr_t)(g.bias + col)) & 15) == 0) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.bias + col), b1 = *reinterpret_cast<const f32x4*>(g.bias + col + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (col + e < g.N) bv[e] = g.bias[col + e];
      }
    }
  }
  if constexpr (SUMOP == 1) {   // D[m][n] = sum_k A[m][k] for every n: take column 0 (lanes 0,16,32,48)
    if (wn == 0 && (lane & 15) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = wrow0 + i * 16 + (lane >> 4) * 4 + r;
          if (row < g.M && sacc[i][r] != 0.f) atomicAdd(g.opsum + row, sacc[i][r]);
        }
    }
  } else if constexpr (SUMOP == 2) {   // D[m][n] = sum_k B[k][n] for every m: take row 0 (lanes 0..15, register 0)
    if (wm == 0 && lane < 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t col = wcol0 + j * 16 + lane;
        if (col < g.N && sacc[j][0] != 0.f) atomicAdd(g.opsum + col, sacc[j][0]);
      }
    }
  }
  // ---- interior wave tiles of bf16 problems (every one at M = B*T rows): branch-free store passes.  What they read from
  // global memory - the ReLU gate's source, the old C when accumulating - is requested for a whole half-tile (4 passes)
  // at once, the gate one half ahead.  In the general passes below every such load is followed by s_waitcnt vmcnt(0),
  // which also waits for the previous pass's store: 16 dependent round trips to HBM per tile (the ffn2 data gradient
  // with its gate took 131 us in the step against 99 us on cache-resident operands).
  // (weight gradients too - fp32 C += - whose eight store passes each waited for the old C behind the previous pass's store)
  if constexpr (sizeof(T) == 2) {
    const bool interior = g.split_k <= 1 && g.c_vec && wrow0 + 64 <= g.M && wcol0 + 64 <= g.N;
    if (__builtin_amdgcn_readfirstlane((int)interior)) {
      const int cc = (lane & 7) * 8;
      const int64_t base = (wrow0 + (lane >> 3)) * g.ldc + wcol0 + cc;     // + (half * 32 + pass * 8) * ldc
      const bf16_t* const rsb = reinterpret_cast<const bf16_t*>(rs);
      const bool gate = EPI == 0 ? rsb != nullptr : EPI == 3, accum = EPI == 0 ? g.accumulate != 0 : EPI == 2;
      bf16x8 pg[2][4];
      f32x4 po[4][sizeof(TC) == 2 ? 1 : 2];
      auto fetch_gate = [&](int half) __attribute__((always_inline)) {
        if (gate) {
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) pg[half][pass] = *reinterpret_cast<const bf16x8*>(rsb + base + (int64_t)(half * 32 + pass * 8) * g.ldc);
        }
      };
      fetch_gate(0);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (accum) {
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) {
            const TC* cp = C + base + (int64_t)(half * 32 + pass * 8) * g.ldc;
            po[pass][0] = *reinterpret_cast<const f32x4*>(cp);             // 8 bf16 or the first 4 floats
            if constexpr (sizeof(TC) == 4) po[pass][1] = *reinterpret_cast<const f32x4*>(cp + 4);
          }
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(ii * 16 + (lane >> 4) * 4 + r) * EPI_LD + j * 16 + (lane & 15)] = acc[half * 2 + ii][j][r];
        if (half == 0) fetch_gate(1);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int rr = pass * 8 + (lane >> 3);
          const f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + rr * EPI_LD + cc);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + rr * EPI_LD + cc + 4);
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = lo[e] + bv[e]; v[4 + e] = hi[e] + bv[4 + e]; }
          if (g.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (gate) {
            const bf16x8 m = pg[half][pass];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)m[e] > 0.f ? v[e] : 0.f;
          }
          TC* cp = C + base + (int64_t)(half * 32 + pass * 8) * g.ldc;
          if constexpr (sizeof(TC) == 2) {
            if (accum) {
              const bf16x8 old = __builtin_bit_cast(bf16x8, po[pass][0]);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += (float)old[e];
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
            STORE_C(reinterpret_cast<bf16x8*>(cp), o);
          } else {
            f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
            if (accum) { o0 += po[pass][0]; o1 += po[pass][1]; }
            STORE_C(reinterpret_cast<f32x4*>(cp), o0);
            STORE_C(reinterpret_cast<f32x4*>(cp + 4), o1);
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cs[(ii * 16 + (lane >> 4) * 4 + r) * EPI_LD + j * 16 + (lane & 15)] = acc[half * 2 + ii][j][r];
    // same wave reads back what it wrote: LDS operations of one wave complete in order
    if (g.split_k > 1) {
      // 64 consecutive fp32 per wave-instruction = 256 contiguous bytes per atomic instruction
      if constexpr (sizeof(TC) == 4) {
        const int64_t col = wcol0 + lane;
        for (int rr = 0; rr < 32; ++rr) {
          const int64_t row = wrow0 + half * 32 + rr;
          if (row < g.M && col < g.N) {
            const float v = Cs[rr * EPI_LD + lane] + bsplit;
            atomicAdd(reinterpret_cast<float*>(C) + row * g.ldc + col, v);
          }
        }
      }
    } else {
      // lane -> (row = pass*8 + lane/8, 8 consecutive columns): one 16-B (bf16) or two 16-B (fp32) stores per lane
      const int cc = (lane & 7) * 8;
      const int64_t col = wcol0 + cc;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int rr = pass * 8 + (lane >> 3);
        const int64_t row = wrow0 + half * 32 + rr;
        if (row >= g.M || col >= g.N) continue;
        float v[8];
        const f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + rr * EPI_LD + cc);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + rr * EPI_LD + cc + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
        const int64_t off = row * g.ldc + col;
        const bool full = g.c_vec && col + 8 <= g.N;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bv[e];
        if (g.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (full) {
          if (rs) {
            const typename Chunk<T>::V m = *reinterpret_cast<const typename Chunk<T>::V*>(rs + off);
            if constexpr (sizeof(T) == 2) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = to_f32<T>(m[e]) > 0.f ? v[e] : 0.f;
            } else {
              const typename Chunk<T>::V m2 = *reinterpret_cast<const typename Chunk<T>::V*>(rs + off + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[e] = to_f32<T>(m[e]) > 0.f ? v[e] : 0.f; v[4 + e] = to_f32<T>(m2[e]) > 0.f ? v[4 + e] : 0.f; }
            }
          }
          if constexpr (sizeof(TC) == 2) {
            if (g.accumulate) {
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
            if (g.accumulate) {
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
            if (g.accumulate) x += to_f32<TC>(C[off + e]);
            C[off + e] = from_f32<TC>(x);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------- persistent variant for grids of many tiles
// The plain kernel above runs its tiles in rounds of two workgroups per CU that start together and therefore reach
// their epilogues together: the chip alternates between MFMA-only and store-only phases (measured on 16000x2304x768:
// 78 us, of which 17 us is the C write-out at HBM speed with idle matrix cores).  Here 2 workgroups per CU stay
// resident and loop over tiles: a tile's stores drain while the next tile is already multiplying, and its first
// operand tiles are requested before the last K-step of the current one.  The tile schedule is static: in every full
// round of gridDim.x tiles each XCD's workgroups (blockIdx & 7) take a contiguous run of the grouped tile order - what
// the hardware dispatcher + remap do for the plain kernel - and the tiles of the last partial round are dealt evenly to
// the XCDs.  (Fetching tiles from per-XCD atomic counters instead cost 12-23 us per launch in contended atomics.)
// bf16 operands, direct-to-LDS staging, whole K-tiles only (K % BK == 0, K / BK >= 3), split_k == 1
template <typename TC, bool TA, bool TB_KCONTIG, bool SIMPLE>
__global__ void __launch_bounds__(NT) gemm_persist_kernel(GemmArgs g) {
  typedef bf16_t T;
  constexpr int BK = (!TA && !TB_KCONTIG) ? 32 : GemmCfg<T>::BK;
  typedef Stager<T, TA, BK> SA;
  typedef Stager<T, !TB_KCONTIG, BK> SB;
  constexpr int STAGE_BYTES = 2 * (SA::TILE + SB::TILE) * (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) char smem[STAGE_BYTES + PEPI_BYTES];
  T* const lds = reinterpret_cast<T*>(smem);
  auto As = [&](int b) -> T* { return lds + b * SA::TILE; };
  auto Bs = [&](int b) -> T* { return lds + 2 * SA::TILE + b * SB::TILE; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float* const Cs = reinterpret_cast<float*>(smem + STAGE_BYTES) + wave * 16 * PEPI_LD;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int G = gridDim.x, per_xcd = G >> 3;                 // G is a multiple of 8
  const int ntiles = g.tiles_m * g.tiles_n;
  const int full_rounds = ntiles / G, rem = ntiles - full_rounds * G;
  const int rq = rem >> 3, rr8 = rem & 7;                    // the last round's tiles: rq (+1 for the first rr8) per XCD
  const int rem_beg = full_rounds * G + (xcd < rr8 ? xcd * (rq + 1) : rr8 * (rq + 1) + (xcd - rr8) * rq);
  const int rem_cnt = xcd < rr8 ? rq + 1 : rq;
  // tile of round r for this workgroup, or -1 when it has none left
  auto tile_at = [&](int r) -> int {
    if (r < full_rounds) return r * G + xcd * per_xcd + loc;
    if (r == full_rounds && loc < rem_cnt) return rem_beg + loc;
    return -1;
  };
  const int nfull = (int)(g.K / BK);
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);
  constexpr int GM = 8;   // grouped order, as in gemm_kernel
  const int per_group = GM * g.tiles_n;
  auto coords = [&](int t, int64_t& m0, int64_t& n0) {
    const int group = t / per_group, first_m = group * GM;
    const int gsz = g.tiles_m - first_m < GM ? g.tiles_m - first_m : GM;
    m0 = (int64_t)(first_m + (t % per_group) % gsz) * BM;
    n0 = (int64_t)((t % per_group) / gsz) * BN;
  };

  int round = 0;
  int tile = tile_at(0);
  int64_t m0 = 0, n0 = 0;
  int cur = 0;
  if (tile >= 0) {
    coords(tile, m0, n0);
    issue_glds<T, TA, BK>(A, g.lda, m0, 0, g.M, As(0), tid);
    issue_glds<T, !TB_KCONTIG, BK>(B, g.ldb, n0, 0, g.N, Bs(0), tid);
  }
  while (tile >= 0) {
    const int next = tile_at(round + 1);
    f32x4 acc[4][4], sacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int64_t nm0 = 0, nn0 = 0;
    // bias of the columns this lane holds in the MFMA layout: requested here, landed by the first barrier (requested
    // inside the K loop they become loop-carried pending loads and the compiler waits for them in every iteration)
    float bcol[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t col = n0 + wn * 64 + j * 16 + (lane & 15);
        bcol[j] = col < g.N ? g.bias[col] : 0.f;
      }
    }
    for (int it = 0; it < nfull; ++it) {
      __syncthreads();   // K-tile `it` has landed (vmcnt(0) precedes the barrier) and buf[cur^1] is no longer read
      if (it + 1 < nfull) {
        const int64_t kn = (int64_t)(it + 1) * BK;
        issue_glds<T, TA, BK>(A, g.lda, m0, kn, g.M, As(cur ^ 1), tid);
        issue_glds<T, !TB_KCONTIG, BK>(B, g.ldb, n0, kn, g.N, Bs(cur ^ 1), tid);
      } else if (next >= 0) {   // first K-tile of the next output tile: in flight during this tile's last K-step and epilogue
        coords(next, nm0, nn0);
        issue_glds<T, TA, BK>(A, g.lda, nm0, 0, g.M, As(cur ^ 1), tid);
        issue_glds<T, !TB_KCONTIG, BK>(B, g.ldb, nn0, 0, g.N, Bs(cur ^ 1), tid);
      }
      mma_tile<T, SA, SB, 0>(As(cur), Bs(cur), acc, sacc, false, wm, wn, lane);
      cur ^= 1;
    }
    persist_epilogue<T, TC, 4, SIMPLE>(g, acc, bcol, Cs, m0 + wm * 64, n0 + wn * 64, lane);
    tile = next; m0 = nm0; n0 = nn0;
    ++round;
  }
}

// ---------------------------------------------------------------- thin-tile variant for the smallest grids
// A decoder-size problem (M = 1280 rows, N = 768) has 60 tiles of 128 x 128: 60 of 256 CUs work, and each of their waves
// issues 64 MFMAs per K-step, so the K loop (12 - 156 steps) is a serial chain of ~0.45 us links.  32 x 128 tiles (wave
// tile 16 x 64) put 240 workgroups on the chip with 16 MFMAs per wave and K-step.  Same 4-stage direct-to-LDS pipeline
// as the small-grid path of gemm_kernel; A is K-contiguous (NT / NN).
// BMT = 64 (wave tile 32 x 64): for grids of 65 - 128 tiles of 128 x 128 (2560 decoder rows at d = 512: 80 tiles -> 160 workgroups)
template <typename TC, bool TB_KCONTIG, bool SIMPLE, int BMT = 32>
__global__ void __launch_bounds__(NT) gemm_thin_kernel(GemmArgs g) {
  typedef bf16_t T;
  constexpr int BK = 64, NS = 4, MI = BMT / 32;
  static_assert(BMT == 32 || BMT == 64, "row tiles of 32 or 64");
  typedef Stager<T, false, BK, BMT> SA;
  typedef Stager<T, !TB_KCONTIG, BK> SB;
  constexpr int STAGE_BYTES = NS * (SA::TILE + SB::TILE) * (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) char smem[STAGE_BYTES + PEPI_BYTES];
  T* const lds = reinterpret_cast<T*>(smem);
  auto As = [&](int b) -> T* { return lds + b * SA::TILE; };
  auto Bs = [&](int b) -> T* { return lds + NS * SA::TILE + b * SB::TILE; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float* const Cs = reinterpret_cast<float*>(smem + STAGE_BYTES) + wave * 16 * PEPI_LD;
  // row tiles fastest: the workgroups that share one 128-column panel of B are neighbours
  const int tm = blockIdx.x % g.tiles_m, tn = blockIdx.x / g.tiles_m;
  const int64_t m0 = (int64_t)tm * BMT, n0 = (int64_t)tn * BN;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);
  f32x4 acc[MI][4], sacc[4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bcol[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t col = n0 + wn * 64 + j * 16 + (lane & 15);
      bcol[j] = col < g.N ? g.bias[col] : 0.f;
    }
  }
  const int nfull = (int)(g.K / BK);
  constexpr int LPT = SA::NCH + SB::NCH;
  static_assert(LPT * (NS - 2) <= 63, "vmcnt field");
  // the bias loads above are older than every tile load: the counted waits below cover them too
  const int npro = nfull < NS - 1 ? nfull : NS - 1;
  for (int t = 0; t < npro; ++t) {
    issue_glds<T, false, BK, BMT>(A, g.lda, m0, (int64_t)t * BK, g.M, As(t), tid);
    issue_glds<T, !TB_KCONTIG, BK>(B, g.ldb, n0, (int64_t)t * BK, g.N, Bs(t), tid);
  }
  int cur = 0;
  for (int it = 0; it < nfull; ++it) {
    const int issued = it + NS - 1 < nfull ? it + NS - 1 : nfull;
    const int inflight = issued - it - 1;          // tiles newer than `it` already issued: 0 .. NS-2
    if (inflight >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT * 2) : "memory");
    else if (inflight == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // every wave's share of tile `it` is in LDS; buffer (it-1)%NS is free
    __builtin_amdgcn_sched_barrier(0);
    if (it + NS - 1 < nfull) {
      const int nb = (cur + NS - 1) % NS;
      const int64_t kn = (int64_t)(it + NS - 1) * BK;
      issue_glds<T, false, BK, BMT>(A, g.lda, m0, kn, g.M, As(nb), tid);
      issue_glds<T, !TB_KCONTIG, BK>(B, g.ldb, n0, kn, g.N, Bs(nb), tid);
    }
    mma_tile<T, SA, SB, 0, MI>(As(cur), Bs(cur), acc, sacc, false, wm, wn, lane);
    cur = (cur + 1) % NS;
  }
  const int64_t ktail = (int64_t)nfull * BK;
  if (ktail < g.K) {   // partial K-tile: register staging with zero fill (as in gemm_kernel)
    SA sa;
    SB sb;
    sa.load(A, g.lda, m0, ktail, g.M, g.K, true, tid);
    sb.load(B, g.ldb, n0, ktail, g.N, g.K, true, tid);
    __syncthreads();
    sa.store(As(cur), tid);
    sb.store(Bs(cur), tid);
    __syncthreads();
    mma_tile<T, SA, SB, 0, MI>(As(cur), Bs(cur), acc, sacc, false, wm, wn, lane);
  }
  persist_epilogue<T, TC, MI, SIMPLE>(g, acc, bcol, Cs, m0 + wm * (16 * MI), n0 + wn * 64, lane);
}

static int thin_max_tiles() {
  static const int v = [] { const char* e = getenv("NEURST_GEMM_THIN_MAX_TILES"); return e ? atoi(e) : 64; }();
  return v;
}
static int thin64_max_tiles() {
  static const int v = [] { const char* e = getenv("NEURST_GEMM_THIN64_MAX_TILES"); return e ? atoi(e) : 128; }();
  return v;
}
// true if the launch was taken by a thin-tile kernel: K-contiguous A, at most thin_max_tiles() tiles of 128 x 128 (row tiles of
// 32) or thin64_max_tiles() of them (row tiles of 64)
template <typename TC, int BMT>
static void launch_thin_bmt(const GemmArgs& g, int trans_b, hipStream_t s) {
  GemmArgs t = g;
  t.tiles_m = (int)cdiv64(g.M, BMT);
  const int grid = t.tiles_m * t.tiles_n;
  const bool simple = !g.relu_src && !g.accumulate && g.c_vec && g.N % 8 == 0;
  if (trans_b) { if (simple) gemm_thin_kernel<TC, true, true, BMT><<<grid, NT, 0, s>>>(t); else gemm_thin_kernel<TC, true, false, BMT><<<grid, NT, 0, s>>>(t); }
  else { if (simple) gemm_thin_kernel<TC, false, true, BMT><<<grid, NT, 0, s>>>(t); else gemm_thin_kernel<TC, false, false, BMT><<<grid, NT, 0, s>>>(t); }
}
template <typename TC>
static bool launch_gemm_thin(const GemmArgs& g, int trans_a, int trans_b, hipStream_t s) {
  if (trans_a || g.split_k != 1 || g.K < 256) return false;
  const int64_t tiles = (int64_t)g.tiles_m * g.tiles_n;
  if (tiles <= thin_max_tiles()) launch_thin_bmt<TC, 32>(g, trans_b, s);
  else if (tiles <= thin64_max_tiles()) launch_thin_bmt<TC, 64>(g, trans_b, s);
  else return false;
  return true;
}

static int persist_min_tiles() {
  static const int v = [] { const char* e = getenv("NEURST_GEMM_PERSIST_MIN_TILES"); return e ? atoi(e) : 1024; }();
  return v;
}

// true if the launch was taken by the persistent kernel.  Measured on the LUT shapes (M = 16000, K = 768): NT 1.5-3 %
// faster than the plain kernel from 1024 tiles up (no gain below: a single round has nothing to overlap); NN (K-tiles of
// 32, twice the barriers per tile) 8 % slower, so only NT takes it.  NEURST_GEMM_PERSIST_MIN_TILES moves the threshold.
template <typename TC>
static bool launch_gemm_persist(const GemmArgs& g, int trans_a, int trans_b, hipStream_t s) {
  if (trans_a || !trans_b) return false;
  const int64_t tiles = (int64_t)g.tiles_m * g.tiles_n;
  if (g.split_k != 1 || g.K % 64 != 0 || g.K / 64 < 3 || tiles < persist_min_tiles()) return false;
  const int grid = 512;   // 2 resident workgroups on each of the 256 CUs; the kernel needs a multiple of 8
  if (!g.relu_src && !g.accumulate && g.c_vec && g.N % 8 == 0) gemm_persist_kernel<TC, false, true, true><<<grid, NT, 0, s>>>(g);
  else gemm_persist_kernel<TC, false, true, false><<<grid, NT, 0, s>>>(g);
  return true;
}

// grids of at most one workgroup per CU cannot hide load latency behind other workgroups: deep pipeline instead
static inline bool small_grid(const GemmArgs& g) { return (int64_t)g.tiles_m * g.tiles_n * g.split_k <= 256 && g.K >= 4 * 64; }

template <typename T, typename TC, bool FAST, int NS>
static int launch_gemm_ns(const GemmArgs& g, int trans_a, int trans_b, hipStream_t s) {
  dim3 grid(g.tiles_m * g.tiles_n, 1, g.split_k);
  if constexpr (FAST && sizeof(T) == 2 && NS == 2) {
    // the large-grid forward / data-gradient calls: the interior passes compiled for exactly what this call asks of them
    if (!trans_a && g.split_k <= 1) {
      const int epi = (!g.relu_src && !g.accumulate) ? 1 : (!g.relu_src ? 2 : (!g.accumulate ? 3 : 0));
      if (trans_b) {
        if (epi == 1) { gemm_kernel<T, TC, false, true, FAST, 0, NS, 1><<<grid, NT, 0, s>>>(g); return 0; }
        if (epi == 2) { gemm_kernel<T, TC, false, true, FAST, 0, NS, 2><<<grid, NT, 0, s>>>(g); return 0; }
        if (epi == 3) { gemm_kernel<T, TC, false, true, FAST, 0, NS, 3><<<grid, NT, 0, s>>>(g); return 0; }
      } else {
        if (epi == 1) { gemm_kernel<T, TC, false, false, FAST, 0, NS, 1><<<grid, NT, 0, s>>>(g); return 0; }
        if (epi == 2) { gemm_kernel<T, TC, false, false, FAST, 0, NS, 2><<<grid, NT, 0, s>>>(g); return 0; }
        if (epi == 3) { gemm_kernel<T, TC, false, false, FAST, 0, NS, 3><<<grid, NT, 0, s>>>(g); return 0; }
      }
    }
  }
  if (!trans_a && trans_b) gemm_kernel<T, TC, false, true, FAST, 0, NS><<<grid, NT, 0, s>>>(g);
  else if (!trans_a && !trans_b) gemm_kernel<T, TC, false, false, FAST, 0, NS><<<grid, NT, 0, s>>>(g);
  else if (trans_a && !trans_b) gemm_kernel<T, TC, true, false, FAST, 0, NS><<<grid, NT, 0, s>>>(g);
  else gemm_kernel<T, TC, true, true, FAST, 0, NS><<<grid, NT, 0, s>>>(g);
  return 0;
}
template <typename T, typename TC, bool FAST>
static int launch_gemm(const GemmArgs& g, int trans_a, int trans_b, hipStream_t s) {
  if constexpr (FAST) {
    if (small_grid(g)) {
      if (launch_gemm_thin<TC>(g, trans_a, trans_b, s)) return 0;
      return launch_gemm_ns<T, TC, true, 4>(g, trans_a, trans_b, s);
    }
    if (launch_gemm_persist<TC>(g, trans_a, trans_b, s)) return 0;
  }
  return launch_gemm_ns<T, TC, FAST, 2>(g, trans_a, trans_b, s);
}

// run-time switches (tests and A/B timing run both kernels in one process); initialised from the environment
static int g_opt_gemm256 = [] { const char* e = getenv("NEURST_GEMM256"); return e ? atoi(e) : 1; }();
int g_opt_gemm256_bn = 0;
int g_opt_gemm256_ablate = 0;
int g_opt_gemm256_max_grid = 0;
int g_opt_gemm256_min_tiles = [] { const char* e = getenv("NEURST_GEMM256_MIN_TILES"); return e ? atoi(e) : 128; }();
static bool use_gemm256() { return g_opt_gemm256 != 0; }
int neurst_attention2_set_option(const char* name, int value);
extern int g_opt_conv_fast;                                  // conv.hip
extern int g_opt_ln_prefetch;                                // layernorm.hip
extern "C" int neurst_set_option(const char* name, int value) {
  NEURST_CHECK_ARG(name, "set_option: null name");
  if (!strcmp(name, "gemm256")) g_opt_gemm256 = value;
  else if (!strcmp(name, "gemm256_min_tiles")) g_opt_gemm256_min_tiles = value;
  else if (!strcmp(name, "gemm256_bn")) g_opt_gemm256_bn = value;
  else if (!strcmp(name, "gemm256_ablate")) g_opt_gemm256_ablate = value;
  else if (!strcmp(name, "gemm256_max_grid")) g_opt_gemm256_max_grid = value;
  else if (!strcmp(name, "conv_fast")) g_opt_conv_fast = value;
  else if (!strcmp(name, "ln_prefetch")) g_opt_ln_prefetch = value;
  else if (neurst_attention2_set_option(name, value)) {}
  else { neurst_set_error("set_option: unknown option '%s'", name); return NEURST_E_INVALID; }
  return NEURST_OK;
}
static int fill_gemm_args(GemmArgs& g, const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                          int64_t ldc, int ab_dtype, int c_dtype) {
  NEURST_CHECK_ARG(A && B && C, "gemm: null operand");
  NEURST_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  size_t es = dtype_size(ab_dtype);
  g.a_aligned = (((uintptr_t)A & 15) == 0) && ((lda * es) % 16 == 0);
  g.b_aligned = (((uintptr_t)B & 15) == 0) && ((ldb * es) % 16 == 0);
  int64_t tm = cdiv64(M, BM), tn = cdiv64(N, BN);
  NEURST_CHECK_ARG(tm * tn < (1ll << 30), "gemm: too many tiles");
  g.tiles_m = (int)tm; g.tiles_n = (int)tn;
  g.c_vec = (((uintptr_t)C & 15) == 0) && ((ldc * dtype_size(c_dtype)) % 16 == 0);
  return NEURST_OK;
}

// Weight gradient + bias gradient of one dense layer in one launch (autodiff of st/models/tools/utils.py:367-398):
//   C[M,N] += A^T . B  with A stored [K][M], B stored [K][N] (K = rows of the layer input), fp32 C, split-K atomics;
//   sum_operand 1: opsum[m] += sum_k A[k][m]  ('out_in' kernels: A is dY);  2: opsum[n] += sum_k B[k][n]  ('in_out': B is dY).
template <typename T, bool FAST>
static void launch_wgrad(const GemmArgs& g, int which, hipStream_t s) {
  dim3 grid(g.tiles_m * g.tiles_n, 1, g.split_k);
  if constexpr (FAST) {
    if (small_grid(g)) {
      if (which == 1) gemm_kernel<T, float, true, false, true, 1, 4><<<grid, NT, 0, s>>>(g);
      else gemm_kernel<T, float, true, false, true, 2, 4><<<grid, NT, 0, s>>>(g);
      return;
    }
  }
  if (which == 1) gemm_kernel<T, float, true, false, FAST, 1><<<grid, NT, 0, s>>>(g);
  else gemm_kernel<T, float, true, false, FAST, 2><<<grid, NT, 0, s>>>(g);
}
extern "C" int neurst_gemm_wgrad_bias(const void* A, const void* B, float* C, float* opsum, int sum_operand, int64_t M, int64_t N,
                                      int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int ab_dtype, int split_k, void* stream) {
  NEURST_CHECK_ARG(opsum && (sum_operand == 1 || sum_operand == 2), "gemm_wgrad_bias: opsum must be given and sum_operand 1 or 2");
  NEURST_CHECK_ARG(lda >= M && ldb >= N && ldc >= N, "gemm_wgrad_bias: leading dimension too small");
  if (split_k < 1) split_k = 1;
  NEURST_CHECK_ARG(split_k <= 64, "gemm: split_k too large");
  GemmArgs g;
  int rc = fill_gemm_args(g, A, B, C, M, N, K, lda, ldb, ldc, ab_dtype, NEURST_F32);
  if (rc) return rc;
  g.accumulate = 1; g.split_k = split_k; g.opsum = opsum;
  hipStream_t s = as_stream(stream);
  if (ab_dtype == NEURST_BF16) {
    const bool fast = g.a_aligned && g.b_aligned && K % 8 == 0 && M % 8 == 0 && N % 8 == 0;
    // long reductions (K = rows of the batch): whole-K 256 x 256 tiles, one workgroup each, no split-K and no atomics on C
    // (split_k is a hint: the sum is the same); few workgroups per launch, meant to run beside other launches
    if (fast && sum_operand == 1 && use_gemm256() && g_opt_gemm256_min_tiles <= 1 && neurst_gemm256_eligible(g, 1, 0, true)) {
      g.split_k = 1;
      neurst_gemm256_launch(g, 1, 0, NEURST_F32, s);
      NEURST_CHECK_LAUNCH("gemm256_wgrad");
      return NEURST_OK;
    }
    if (fast) launch_wgrad<bf16_t, true>(g, sum_operand, s); else launch_wgrad<bf16_t, false>(g, sum_operand, s);
  } else {
    launch_wgrad<float, false>(g, sum_operand, s);
  }
  NEURST_CHECK_LAUNCH("gemm_wgrad_bias");
  return NEURST_OK;
}

extern "C" int neurst_gemm_bias_act(const void* A, const void* B, void* C, const float* bias, const void* relu_src, int64_t M,
                                    int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int trans_a, int trans_b,
                                    int ab_dtype, int c_dtype, int relu, int accumulate, int split_k, void* stream);
// Several weight (+ bias) gradients of `dense` layers in one launch (autodiff of st/models/tools/utils.py:367-398 for the
// layers of one or two Transformer blocks): problem i is C_i[M_i,N_i] += A_i^T . B_i, A_i [K_i,M_i] = dY, B_i [K_i,N_i] = X,
// opsum_i (may be NULL) += column sums of A_i.  bf16 operands, fp32 C.
extern "C" int neurst_gemm_wgrad_bias_grouped_tiles(int n, const void* const* A, const void* const* B, float* const* C, float* const* opsum,
                                                    const int64_t* M, const int64_t* N, const int64_t* K, const int64_t* lda, const int64_t* ldb,
                                                    const int64_t* ldc, const int* tile_begin, const int* tile_end, int ab_dtype, void* stream) {
  constexpr int MAXP = 12;
  NEURST_CHECK_ARG(n >= 1 && n <= MAXP && n <= neurst_gemm256_group_max() && A && B && C && opsum && M && N && K && lda && ldb && ldc,
                   "gemm_wgrad_bias_grouped: 1 to %d problems", MAXP);
  NEURST_CHECK_ARG((tile_begin == nullptr) == (tile_end == nullptr), "gemm_wgrad_bias_grouped: tile_begin and tile_end go together");
  GemmArgs gs[MAXP];
  for (int i = 0; i < n; ++i) {
    NEURST_CHECK_ARG(lda[i] >= M[i] && ldb[i] >= N[i] && ldc[i] >= N[i], "gemm_wgrad_bias_grouped: leading dimension too small (problem %d)", i);
    if (tile_begin)
      NEURST_CHECK_ARG(tile_begin[i] >= 0 && tile_begin[i] <= tile_end[i] && tile_end[i] <= neurst_gemm256_group_tiles(M[i], N[i]),
                       "gemm_wgrad_bias_grouped: tile range of problem %d outside 0..%lld", i, (long long)neurst_gemm256_group_tiles(M[i], N[i]));
    int rc = fill_gemm_args(gs[i], A[i], B[i], C[i], M[i], N[i], K[i], lda[i], ldb[i], ldc[i], ab_dtype, NEURST_F32);
    if (rc) return rc;
    gs[i].accumulate = 1; gs[i].split_k = 1; gs[i].opsum = opsum[i];
  }
  hipStream_t s = as_stream(stream);
  // Eligibility for the grouped kernel is a PER-PROBLEM predicate (alignment, M % 8 / N % 8, 32-bit window, tile counts): a
  // gradient whose tiles straddle two calls takes the same path in both, whatever else is queued beside it.  Eligible
  // problems go out together in one grouped launch; an ineligible one is computed WHOLE by the call that holds its tile 0.
  int ne = 0, emap[MAXP];
  GemmArgs ge[MAXP];
  int eb[MAXP], ee[MAXP];
  for (int i = 0; i < n; ++i) {
    if (!(ab_dtype == NEURST_BF16 && use_gemm256() && neurst_gemm256_group_eligible(gs[i]))) continue;
    emap[ne] = i;
    ge[ne] = gs[i];
    eb[ne] = tile_begin ? tile_begin[i] : 0;
    ee[ne] = tile_end ? tile_end[i] : (int)neurst_gemm256_group_tiles(M[i], N[i]);
    ++ne;
  }
  bool grouped[MAXP] = {};
  if (ne > 0) {
    if (!neurst_gemm256_group_launch(ne, ge, eb, ee, s)) {
      neurst_set_error("gemm_wgrad_bias_grouped: grouped launch refused an eligible problem");
      return NEURST_E_INVALID;
    }
    NEURST_CHECK_LAUNCH("gemm256_group");
    for (int e = 0; e < ne; ++e) grouped[emap[e]] = true;
  }
  for (int i = 0; i < n; ++i) {                                // one launch per gradient (fp32 parity mode, odd shapes):
    if (grouped[i]) continue;
    if (tile_begin && (tile_begin[i] != 0 || tile_end[i] == 0)) continue;   // whole, by the call that holds its first tile
    int rc;
    if (opsum[i]) rc = neurst_gemm_wgrad_bias(A[i], B[i], C[i], opsum[i], 1, M[i], N[i], K[i], lda[i], ldb[i], ldc[i], ab_dtype, 1, stream);
    else rc = neurst_gemm_bias_act(A[i], B[i], C[i], nullptr, nullptr, M[i], N[i], K[i], lda[i], ldb[i], ldc[i], 1, 0, ab_dtype, NEURST_F32, 0, 1, 1, stream);
    if (rc) return rc;
  }
  return NEURST_OK;
}
extern "C" int neurst_gemm_wgrad_bias_grouped(int n, const void* const* A, const void* const* B, float* const* C, float* const* opsum,
                                              const int64_t* M, const int64_t* N, const int64_t* K, const int64_t* lda, const int64_t* ldb,
                                              const int64_t* ldc, int ab_dtype, void* stream) {
  NEURST_CHECK_ARG(n >= 1 && n <= 8, "gemm_wgrad_bias_grouped: 1 to 8 problems");
  return neurst_gemm_wgrad_bias_grouped_tiles(n, A, B, C, opsum, M, N, K, lda, ldb, ldc, nullptr, nullptr, ab_dtype, stream);
}
extern "C" int64_t neurst_gemm_wgrad_group_tiles(int64_t M, int64_t N) { return neurst_gemm256_group_tiles(M, N); }
extern "C" int neurst_gemm_wgrad_group_max(void) { return neurst_gemm256_group_max(); }

extern "C" int neurst_gemm_bias_act(const void* A, const void* B, void* C, const float* bias, const void* relu_src, int64_t M,
                                    int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int trans_a, int trans_b,
                                    int ab_dtype, int c_dtype, int relu, int accumulate, int split_k, void* stream) {
  NEURST_CHECK_ARG(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, "gemm: leading dimension too small");
  NEURST_CHECK_ARG(!(ab_dtype == NEURST_F32 && c_dtype == NEURST_BF16), "gemm: fp32 operands need an fp32 result");
  if (split_k < 1) split_k = 1;
  NEURST_CHECK_ARG(split_k == 1 || (c_dtype == NEURST_F32 && accumulate && !relu && !relu_src),
                   "gemm: split_k > 1 needs fp32 C, accumulate=1 and no activation");
  NEURST_CHECK_ARG(split_k <= 64, "gemm: split_k too large");
  GemmArgs g;
  int rc = fill_gemm_args(g, A, B, C, M, N, K, lda, ldb, ldc, ab_dtype, c_dtype);
  if (rc) return rc;
  g.bias = bias; g.relu_src = relu_src;
  g.relu = relu; g.accumulate = accumulate; g.split_k = split_k;
  const size_t es = dtype_size(ab_dtype);
  if (relu_src && !((((uintptr_t)relu_src & 15) == 0) && ((ldc * es) % 16 == 0))) g.c_vec = 0;
  hipStream_t s = as_stream(stream);
  if (ab_dtype == NEURST_BF16) {
    // FAST: direct-to-LDS staging needs 16-byte aligned rows and whole 8-element chunks along every contiguous axis
    const bool fast = g.a_aligned && g.b_aligned && K % 8 == 0 && (!trans_a || M % 8 == 0) && (trans_b || N % 8 == 0);
    if (fast && use_gemm256() && neurst_gemm256_eligible(g, trans_a, trans_b, c_dtype == NEURST_F32)) {
      g.split_k = 1;                                           // a hint only: C += A . B is the same sum
      neurst_gemm256_launch(g, trans_a, trans_b, c_dtype, s);
      NEURST_CHECK_LAUNCH("gemm256");
      return NEURST_OK;
    }
    if (c_dtype == NEURST_BF16) { if (fast) launch_gemm<bf16_t, bf16_t, true>(g, trans_a, trans_b, s); else launch_gemm<bf16_t, bf16_t, false>(g, trans_a, trans_b, s); }
    else { if (fast) launch_gemm<bf16_t, float, true>(g, trans_a, trans_b, s); else launch_gemm<bf16_t, float, false>(g, trans_a, trans_b, s); }
  } else {
    launch_gemm<float, float, false>(g, trans_a, trans_b, s);
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
