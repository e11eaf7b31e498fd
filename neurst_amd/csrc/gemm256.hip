// K3, large grids: C[M,N] = op(A) . op(B) (+bias)(relu)(relu-grad gate)(+C) for bf16 operands on a 256 x 256 output
// tile per workgroup - the GEMMs with M = B*T rows of `dense` (st/models/tools/utils.py:367-398), tf.layers.dense and
// their data gradients, where the 128 x 128 kernels of gemm.hip reach 25-35 % of the matrix peak.
//
// Structure (cdna_hip_programming.md section 5, "The 256^2 8-phase template", rebuilt here as a ring of operand PIECES):
//  * 8 waves (2 along M x 4 along N), wave tile 128 x 64 = 8 x 4 MFMA tiles of v_mfma_f32_16x16x32_bf16 (128 accumulator
//    registers), K-tile 64.  One workgroup per CU (144 KiB LDS), persistent: it walks its output tiles and keeps the
//    operand stream running across tile seams (the next tile's first pieces are in flight during the epilogue).
//  * A K-tile is cut into 4 pieces of 16 KiB, each read from LDS in exactly ONE phase:
//        piece 0: the first 32 of every wave column's 64 B-columns   (read in phase 0, kept in registers until phase 3)
//        piece 1: the first 64 of every wave row's 128 A-rows         (read in phase 0)
//        piece 2: the other 32 B-columns                              (read in phase 1)
//        piece 3: the other 64 A-rows                                 (read in phase 2)
//    phase 0: C[m0][n0] += A0.B0, phase 1: C[m0][n1] += A0.B1, phase 2: C[m1][n1] += A1.B1, phase 3: C[m1][n0] += A1.B0
//    (16 MFMAs per wave and phase).  Pieces are numbered s = 4 * (K-tile) + piece and live in LDS slot s mod 8.
//  * Every phase issues ONE piece, 6 pieces ahead of the one numbered like the phase (2 buffer_load ... lds per thread),
//    and waits with a COUNTED vmcnt for the pieces the next phase reads: four pieces (64 KiB per CU) stay in flight
//    across the barriers.  Slot reuse: piece s+8 is issued in phase s+2, at least two phases after the last read of
//    piece s; a piece is read one phase after the wait + barrier that retires it (both groups of waves, see below).
//  * The waves of wave-row 1 run one barrier behind those of wave-row 0 (waves w and w+4 share a SIMD): while one of the
//    two computes its 16 MFMAs the other reads LDS and issues loads, two barriers per phase.
//  * Operands arrive by buffer_load ... lds with the XOR swizzle on the SOURCE side (rule 21); rows / columns beyond the
//    operand's end read as zero through the descriptor's range check, so there is no clamping in the loop.  k-strided
//    operands ([K][rows] in memory: the weight of a data gradient, both operands of a weight gradient) keep their memory
//    layout in LDS and are read with the transposing ds_read_b64_tr_b16.
#include <type_traits>

#include "common.h"
#include "gemm_tiles.h"

namespace t2 {

constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
constexpr int PIECE = 128 * 64;                                // elements of one piece (16 KiB)
constexpr int RING_BYTES = 8 * PIECE * 2;                      // 128 KiB
constexpr int EPI_BYTES = 8 * 16 * PEPI_LD * 4;                // one 16 x 32 fp32 staging tile per wave
constexpr int OOB = 0x7ffffff0;                                // byte offset beyond any descriptor's range
constexpr int64_t MAX_RECORDS = 0x7fff0000;

typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((address_space(3))) char* lds_cp;

// raw buffer descriptor (stride 0, 32-bit records in bytes, gfx9 raw-buffer flags): built by hand so that it can be an
// SGPR operand of the asm statement below
__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned records) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, records, 0x00020000u};
}
// One LDS-DMA wave-instruction: 64 lanes x 16 bytes from rsrc + voff (per lane) to LDS bytes lds_addr .. lds_addr + 1023
// (wave-uniform, lane-linear).  In inline asm on purpose: hipcc then neither counts it in its own vmcnt bookkeeping nor
// orders LDS reads behind it - with the builtin it drains the whole ring (vmcnt(0)) at the head of every K-loop iteration.
// Completion is counted by hand (the vmcnt(8) waits of the K loop).  M0 is saved and restored inside the statement
// (cdna_hip_programming.md 5.7); the s_nop covers M0 -> LDS-DMA and a freshly written descriptor register.
__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned lds_addr, int voff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
typedef Stager<bf16_t, false, 64, 128> SK;                     // K-contiguous piece: [128 rows][64 k], 128-byte rows
typedef Stager<bf16_t, true, 64> ST;                           // k-strided piece:    [64 k][128 columns], 256-byte rows

// Byte offset, relative to the operand's (tile, K-tile) origin, of the 16-byte chunk this thread fetches in issue i (0, 1)
// of the piece holding half `sub` (0, 1) of every wave row's A rows (IS_A) or of every wave column's B columns.
// WN = columns per wave (64 or 48); the second B piece then holds 32 or 16 columns per wave.  Returns OOB for the
// positions of an LDS image that no operand element belongs to (k-strided second B piece of a 48-column wave tile).
template <bool KSTRIDED, bool IS_A, int WN>
__device__ __forceinline__ int piece_voff(int tid, int i, int sub, int64_t ld) {
  constexpr int W1 = WN - 32;                                  // columns per wave in the second B piece: 32 or 16
  if constexpr (!KSTRIDED) {
    const int r = i * 64 + (tid >> 3);                         // row of the LDS image
    const int c = (tid & 7) ^ (r & 7);                         // source chunk that belongs at LDS chunk position tid & 7
    int row;
    if constexpr (IS_A) row = (r >> 6) * 128 + sub * 64 + (r & 63);
    else if (sub == 0) row = (r >> 5) * WN + (r & 31);
    else row = (r / W1) * WN + 32 + (r % W1);
    return (int)(row * ld * 2) + c * 16;
  } else {
    const int P = i * 512 + tid;
    const int r = P >> 4;                                      // k row of the LDS image
    const int c = (P & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
    int col;
    if constexpr (IS_A) col = (c >> 3) * 128 + sub * 64 + (c & 7) * 8;
    else if (sub == 0) col = (c >> 2) * WN + (c & 3) * 8;
    else {
      constexpr int CW = W1 / 8;                               // chunks per wave: 4 or 2
      if (c >= 4 * CW) return OOB;
      col = (c / CW) * WN + 32 + (c % CW) * 8;
    }
    return (int)(r * ld * 2) + col * 2;
  }
}

struct Args {
  GemmArgs g;
  int nk;                 // K-tiles per output tile (even)
  unsigned long long* stamps;   // diagnostic instantiation only: [workgroup][24] clock readings
  int ablate;                   // diagnostic instantiation only: timing-only ablations of the K loop (wrong results)
};
__device__ __forceinline__ unsigned long long clk() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

template <bool KSTRIDED>
__device__ __forceinline__ bf16x8 frag(const bf16_t* __restrict__ piece, int first, int ks, int lane) {
  if constexpr (!KSTRIDED) return SK::frag(piece, first, ks, lane);
  else return ST::frag(piece, first, ks, lane);
}

// Accumulators of one wave (8 x NI MFMA tiles = 128 rows x 16 NI columns) -> C, with bias / ReLU / ReLU-gradient gate /
// accumulate.  Per 16 rows and 32 columns: MFMA layout -> this wave's LDS staging tile -> 8 consecutive columns per lane,
// so that every lane stores 16 bytes (bf16) or 2 x 16 bytes (fp32) of one row.
// EPI (what the host guarantees, so that the unrolled chunks are straight-line code - a run-time flag in them costs 2-5 % of a
// K = 768 problem): 0 nothing; 1 no gate, no accumulation, whole 16-byte chunks (relu_src == NULL, accumulate == 0, c_vec,
// N % 8 == 0: the forward call of a dense layer); 2 C += with whole chunks and no gate (weight gradients): the old C of four
// row groups is in flight ahead of the chunk that adds it, instead of one load - wait - store round trip per chunk.
template <typename TC, int NI, int EPI>
__device__ __forceinline__ void epilogue(const GemmArgs& g, f32x4 (&acc)[8][NI], const float (&bcol)[NI], float* __restrict__ Cs,
                                         int64_t wrow0, int64_t wcol0, int lane) {
  TC* C = reinterpret_cast<TC*>(g.C);
  const bf16_t* rs = EPI ? nullptr : reinterpret_cast<const bf16_t*>(g.relu_src);
  const bool accumulate = EPI ? EPI == 2 : g.accumulate != 0;
  const int lg = lane >> 4, l15 = lane & 15;
  const int rr = lane >> 2, cc = (lane & 3) * 8;
  constexpr int NH = (NI + 1) / 2;                             // 32-column halves (the last one may hold 16 columns)
  constexpr int DEPTH = 4, NV = sizeof(TC) == 4 ? 2 : 1;
  f32x4 oldc[EPI == 2 ? DEPTH : 1][NH][NV];
  auto fetch_old = [&](int i) __attribute__((always_inline)) {
    if constexpr (EPI == 2) {
#pragma unroll
      for (int jh = 0; jh < NH; ++jh) {
        const int nj = NI - 2 * jh < 2 ? NI - 2 * jh : 2;
        const int64_t row = wrow0 + i * 16 + rr, col = wcol0 + jh * 32 + cc;
        if (cc >= nj * 16 || row >= g.M || col >= g.N) continue;
        const TC* cp = C + row * g.ldc + col;
        oldc[i % DEPTH][jh][0] = *reinterpret_cast<const f32x4*>(cp);     // 8 bf16 or the first 4 floats
        if constexpr (NV == 2) oldc[i % DEPTH][jh][1] = *reinterpret_cast<const f32x4*>(cp + 4);
      }
    }
  };
  if constexpr (EPI == 2) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) fetch_old(i);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int jh = 0; jh < NH; ++jh) {
      const int nj = NI - 2 * jh < 2 ? NI - 2 * jh : 2;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        if (jj < nj) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[i][jh * 2 + jj][r] + bcol[jh * 2 + jj];
            if (g.relu) v = fmaxf(v, 0.f);
            Cs[(lg * 4 + r) * PEPI_LD + jj * 16 + l15] = v;
          }
        }
      // the same wave reads back what it wrote: LDS operations of one wave complete in order
      const f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + rr * PEPI_LD + cc);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + rr * PEPI_LD + cc + 4);
      const int64_t row = wrow0 + i * 16 + rr, col = wcol0 + jh * 32 + cc;
      if (cc >= nj * 16 || row >= g.M || col >= g.N) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
      const int64_t off = row * g.ldc + col;
      if (EPI || (g.c_vec && col + 8 <= g.N)) {
        if (rs) {
          const bf16x8 m = *reinterpret_cast<const bf16x8*>(rs + off);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (float)m[e] > 0.f ? v[e] : 0.f;
        }
        if constexpr (sizeof(TC) == 2) {
          if (accumulate) {
            const bf16x8 old = EPI == 2 ? __builtin_bit_cast(bf16x8, oldc[i % DEPTH][jh][0]) : *reinterpret_cast<const bf16x8*>(C + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)old[e];
          }
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
          *reinterpret_cast<bf16x8*>(C + off) = o;
        } else {
          f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
          if constexpr (EPI == 2) {
            o0 += oldc[i % DEPTH][jh][0];
            o1 += oldc[i % DEPTH][jh][NV - 1];
          } else if (accumulate) {
            o0 += *reinterpret_cast<const f32x4*>(C + off);
            o1 += *reinterpret_cast<const f32x4*>(C + off + 4);
          }
          *reinterpret_cast<f32x4*>(C + off) = o0;
          *reinterpret_cast<f32x4*>(C + off + 4) = o1;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (col + e >= g.N) continue;
          float x = v[e];
          if (rs) x = (float)rs[off + e] > 0.f ? x : 0.f;
          if (accumulate) x += to_f32<TC>(C[off + e]);
          C[off + e] = from_f32<TC>(x);
        }
      }
    }
    if (i + DEPTH < 8) fetch_old(i + DEPTH);                   // into the slot row group i has just released
  }
}

// NI1 = MFMA tiles (16 columns each) per wave in the SECOND B piece: 2 -> 256-column output tiles, 1 -> 192-column tiles
// (wave tile 128 x 48; M = 16000 rows give 63 tile rows, so tile counts that are multiples of 4 per tile row fill 256 CUs
// in whole rounds: N = 768 -> 252 tiles, 2304 -> 756, 1536 -> 504).
// ONE: the workgroup computes exactly one output tile, `block` = tile row | tile column << 16 (grouped weight gradients:
// the launch's own placement rule decides which tile a workgroup gets); otherwise the persistent static schedule below.
template <typename TC, bool TA, bool TBK, int NI1, bool STAMPS, int ABL, bool SUMA, bool ONE = false, int EPI = 0>
__device__ __forceinline__ void body(const GemmArgs& g, const int nk, const int block, const int G, unsigned long long* const stamps_buf) {
  constexpr bool AKS = TA, BKS = !TBK;                         // operand is k-strided in memory
  constexpr int NI = 2 + NI1, WN = 16 * NI, BNT = 4 * WN;      // MFMA tiles / columns per wave, columns per output tile
  // loads per thread of the four piece kinds (a K-contiguous second B piece of 64 rows is ONE issue)
  constexpr int NL2 = (NI1 == 1 && !BKS) ? 1 : 2;
  constexpr int NLOAD[4] = {2, 2, NL2, 2};
  __shared__ __attribute__((aligned(16))) char smem[RING_BYTES + EPI_BYTES];   // ONE LDS object
  bf16_t* const ring = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  float* const Cs = reinterpret_cast<float*>(smem + RING_BYTES) + wave * 16 * PEPI_LD;
  const bf16_t* const A = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* const B = reinterpret_cast<const bf16_t*>(g.B);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_cp)smem;      // LDS byte address of the ring

  // ---- static tile schedule (as gemm_persist_kernel): in every full round each XCD's workgroups take a contiguous run of
  // the grouped tile order, the last partial round is dealt evenly to the XCDs
  const int xcd = block & 7, loc = block >> 3;
  const int per_xcd = G >> 3;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int full_rounds = ntiles / G, rem = ntiles - full_rounds * G;
  const int rq = rem >> 3, rr8 = rem & 7;
  const int rem_beg = full_rounds * G + (xcd < rr8 ? xcd * (rq + 1) : rr8 * (rq + 1) + (xcd - rr8) * rq);
  const int rem_cnt = xcd < rr8 ? rq + 1 : rq;
  auto tile_at = [&](int r) -> int {
    if constexpr (ONE) return r == 0 ? block : -1;
    if (r < full_rounds) return r * G + xcd * per_xcd + loc;
    if (r == full_rounds && loc < rem_cnt) return rem_beg + loc;
    return -1;
  };
  constexpr int GM = 8;
  const int per_group = GM * g.tiles_n;
  auto coords = [&](int t, int64_t& m0, int64_t& n0) {
    if constexpr (ONE) { m0 = (int64_t)(t & 0xffff) * BM; n0 = (int64_t)(t >> 16) * BNT; return; }
    const int group = t / per_group, first_m = group * GM;
    const int gsz = g.tiles_m - first_m < GM ? g.tiles_m - first_m : GM;
    m0 = (int64_t)(first_m + (t % per_group) % gsz) * BM;
    n0 = (int64_t)((t % per_group) / gsz) * BNT;
  };
  if (tile_at(0) < 0) return;                                  // whole workgroup: nothing to do
  unsigned long long* const stamp = STAMPS ? stamps_buf + (size_t)block * 24 : nullptr;
  auto put = [&](int i) __attribute__((always_inline)) {
    if constexpr (STAMPS) {
      if (!stamps_buf) return;
      const unsigned long long t = clk();
      if (tid == 0 && i < 24) stamp[i] = t;
    }
  };
  if constexpr (STAMPS) { if (tid == 0 && stamps_buf) stamp[23] = __builtin_amdgcn_s_memrealtime(); }
  put(0);

  // ---- per-thread source offsets of the four piece kinds
  int voA[2][2], voB[2][2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      voA[sub][i] = piece_voff<AKS, true, WN>(tid, i, sub, g.lda);
      voB[sub][i] = piece_voff<BKS, false, WN>(tid, i, sub, g.ldb);
    }
  const int kc8 = ((tid & 7) ^ ((tid >> 3) & 7)) * 8;          // first k of this thread's chunk in a K-contiguous piece

  // ---- load cursor: runs 6 pieces ahead of the compute cursor, across output tiles
  int l_round = 0, l_tile = tile_at(0), l_kt = 0;
  int64_t l_m0 = 0, l_n0 = 0;
  coords(l_tile, l_m0, l_n0);
  u32x4 rsA, rsB;
  int l_krem = 0;
  auto set_rsrc = [&]() __attribute__((always_inline)) {
    if (l_tile < 0) {                                          // past the last tile: every load is out of range (zeros, no traffic)
      rsA = make_rsrc(A, 0);
      rsB = make_rsrc(B, 0);
      l_krem = BK;
      return;
    }
    const int64_t k0 = (int64_t)l_kt * BK;
    if (k0 >= g.K) {                                           // the all-zero K-tile that pads an odd count
      rsA = make_rsrc(A, 0);
      rsB = make_rsrc(B, 0);
      l_krem = BK;
      return;
    }
    // first element of this (tile, K-tile) and one past the operand's last element
    const int64_t a_org = AKS ? k0 * g.lda + l_m0 : l_m0 * g.lda + k0;
    const int64_t a_end = AKS ? (g.K - 1) * g.lda + g.M : (g.M - 1) * g.lda + g.K;
    const int64_t b_org = BKS ? k0 * g.ldb + l_n0 : l_n0 * g.ldb + k0;
    const int64_t b_end = BKS ? (g.K - 1) * g.ldb + g.N : (g.N - 1) * g.ldb + g.K;
    int64_t ra = (a_end - a_org) * 2, rb = (b_end - b_org) * 2;
    ra = ra < 0 ? 0 : (ra > MAX_RECORDS ? MAX_RECORDS : ra);
    rb = rb < 0 ? 0 : (rb > MAX_RECORDS ? MAX_RECORDS : rb);
    rsA = make_rsrc(A + a_org, (unsigned)ra);
    rsB = make_rsrc(B + b_org, (unsigned)rb);
    l_krem = (int)(g.K - k0);
  };
  auto advance_k = [&]() __attribute__((always_inline)) {
    if (l_tile >= 0 && ++l_kt == nk) {
      l_kt = 0;
      l_tile = tile_at(++l_round);
      if (l_tile >= 0) coords(l_tile, l_m0, l_n0);
    }
    set_rsrc();
  };
  set_rsrc();
  // issue i (0, 1) of piece kind 0: B sub 0, 1: A sub 0, 2: B sub 1, 3: A sub 1 of the cursor's K-tile, into LDS slot `slot`
  constexpr int abl = ABL;                                     // timing-only ablations (diagnostic instantiations)
  // The two DMA instructions of a phase's piece are issued in its LOAD segment, one before and one after the LDS reads:
  // each costs the issuing wave 60-180 cycles (the per-CU load path takes ~33 cycles per 1-KiB instruction and is as busy
  // as the matrix pipe: 64 KiB against 8.4 MFLOP per K-tile), which the partner wave's MFMAs cover.  Issued between the
  // MFMAs instead (ABL bit 64 of the diagnostic instantiations) the K loop measured 9 % slower.
  constexpr bool DMA_IN_LOAD = (ABL & 64) == 0;
  auto issue1 = [&](int kind, int slot, int i) __attribute__((always_inline)) {
    if (i >= NLOAD[kind]) return;
    if constexpr ((abl & 1) != 0) return;
    const bool isA = kind & 1;
    const int sub = kind >> 1;
    const unsigned dst = lds0 + (unsigned)(slot * PIECE + wave * 512) * 2;   // this wave's 1 KiB of each 8 KiB issue
    int vo = isA ? voA[sub][i] : voB[sub][i];
    // partial last K-tile of a K-contiguous operand: chunks at or beyond K must read as zero (a k-strided operand gets
    // that from the range check: rows beyond K lie beyond its end)
    if (l_krem < BK && !(isA ? AKS : BKS)) vo = kc8 < l_krem ? vo : OOB;
    dma16(isA ? rsA : rsB, dst + i * 8192, vo);
  };
  auto issue = [&](int kind, int slot) __attribute__((always_inline)) { issue1(kind, slot, 0); issue1(kind, slot, 1); };

  // ---- prologue: pieces 0..5
  issue(0, 0); issue(1, 1); issue(2, 2); issue(3, 3);
  advance_k();
  issue(0, 4); issue(1, 5);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + NL2) : "memory");   // pieces 0 and 1 of this wave have landed
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();                   // wave row 1 runs one barrier behind wave row 0
  __builtin_amdgcn_sched_barrier(0);

  put(1);
  f32x4 acc[8][NI];
  bf16x8 af[4][2], b0[2][2], b1[NI1][2];
  // SUMA (weight gradients): opsum[m] += sum_k A[k][m], the bias gradient of the layer.  The A fragments of a duty K-tile
  // are multiplied once more, as the B operand, against a selector whose only non-zero row is the fragment's own index:
  // ONE accumulator tile collects the column sums of all 8 fragments (row t = sums of fragment t).  K-tiles are dealt
  // round-robin to the 4 wave columns and to the tiles of a tile row, so every A element is summed exactly once.
  f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
  bool duty = false;

  // 4 x NB x 2 MFMAs of one phase with the two DMA issues of piece `kind` spread among them: the issue of an LDS-DMA
  // instruction takes 60-180 cycles of the wave, which are hidden here behind the matrix pipe instead of lengthening the
  // load segment that the partner wave on the SIMD waits for at the next barrier
  auto compute = [&](auto mo_c, auto no_c, auto nb_c, bf16x8 (&bf)[decltype(nb_c)::value][2], int kind, int slot) __attribute__((always_inline)) {
    constexpr int MO = decltype(mo_c)::value, NO = decltype(no_c)::value, NB = decltype(nb_c)::value;
    if constexpr ((abl & 16) == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NB; ++ni)
          if constexpr ((abl & 4) == 0)
            acc[MO + mi][NO + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][ks], bf[ni][ks], acc[MO + mi][NO + ni], 0, 0, 0);
          else asm volatile("" ::"v"(af[mi][ks]), "v"(bf[ni][ks]));   // keeps the LDS reads alive
        if constexpr (!DMA_IN_LOAD) {
          if (mi == 1) {
            __builtin_amdgcn_sched_barrier(0);
            issue1(kind, slot, ks);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    if constexpr (SUMA && NO == 0) {                           // phases 0 and 3: each half of the A rows exactly once
      if (duty) {
        const unsigned one2 = 0x3F803F80u;                     // two bf16 ones
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const unsigned w = (lane & 15) == MO + mi ? one2 : 0u;
          const u32x4 selw = {w, w, w, w};
          const bf16x8 sel = __builtin_bit_cast(bf16x8, selw);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel, af[mi][ks], sacc, 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I2 = std::integral_constant<int, 2>;
  using I4 = std::integral_constant<int, 4>;
  using IN1 = std::integral_constant<int, NI1>;
#define T2_RD(x) do { if constexpr ((abl & 2) == 0) { x; } } while (0)
#define T2_BAR2() do { if constexpr ((abl & 8) == 0) __builtin_amdgcn_s_barrier(); } while (0)
#define T2_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

  // one K-tile = 4 phases; PAR = parity of the K-tile's global index (LDS slots 4 PAR .. 4 PAR + 3).  Phase p issues
  // piece p + 6 during its MFMAs; before its first barrier it waits for the pieces phase p + 1 reads (up to p + 2), with
  // pieces p + 3 .. p + 5 still in flight.
  auto ktile = [&](auto par_c) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    const bf16_t* const s0 = ring + (4 * PAR + 0) * PIECE;
    const bf16_t* const s1 = ring + (4 * PAR + 1) * PIECE;
    const bf16_t* const s2 = ring + (4 * PAR + 2) * PIECE;
    const bf16_t* const s3 = ring + (4 * PAR + 3) * PIECE;
    // ---------------- phase 0: newer than piece p + 2 are kinds 3, 0, 1
    if constexpr (DMA_IN_LOAD) { issue1(2, (4 * PAR + 6) & 7, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) T2_RD(b0[ni][ks] = frag<BKS>(s0, wc * 32 + ni * 16, ks * 32, lane));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) T2_RD(af[mi][ks] = frag<AKS>(s1, wr * 64 + mi * 16, ks * 32, lane));
    if constexpr (DMA_IN_LOAD) { __builtin_amdgcn_sched_barrier(0); issue1(2, (4 * PAR + 6) & 7, 1); T2_WAIT(6 + NL2); } else
    T2_WAIT(6);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    compute(I0{}, I0{}, I2{}, b0, 2, (4 * PAR + 6) & 7);
    T2_BAR2();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- phase 1: newer are kinds 0, 1, 2
#pragma unroll
    for (int ni = 0; ni < NI1; ++ni)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) T2_RD(b1[ni][ks] = frag<BKS>(s2, wc * (16 * NI1) + ni * 16, ks * 32, lane));
    if constexpr (DMA_IN_LOAD) { __builtin_amdgcn_sched_barrier(0); issue(3, (4 * PAR + 7) & 7); T2_WAIT(6 + NL2); } else
    T2_WAIT(4 + NL2);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    compute(I0{}, I2{}, IN1{}, b1, 3, (4 * PAR + 7) & 7);
    advance_k();
    T2_BAR2();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- phase 2: phase 3 reads nothing, no wait
    if constexpr (DMA_IN_LOAD) { issue1(0, 4 * PAR + 0, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) T2_RD(af[mi][ks] = frag<AKS>(s3, wr * 64 + mi * 16, ks * 32, lane));
    if constexpr (DMA_IN_LOAD) { __builtin_amdgcn_sched_barrier(0); issue1(0, 4 * PAR + 0, 1); }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    compute(I4{}, I2{}, IN1{}, b1, 0, 4 * PAR + 0);
    T2_BAR2();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- phase 3: newer are kinds 2, 3, 0
    if constexpr (DMA_IN_LOAD) { issue(1, 4 * PAR + 1); T2_WAIT(6 + NL2); } else
    T2_WAIT(4 + NL2);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    compute(I4{}, I0{}, I2{}, b0, 1, 4 * PAR + 1);
    T2_BAR2();
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int round = 0;; ++round) {
    const int tile = tile_at(round);
    if (tile < 0) break;
    int64_t m0, n0;
    coords(tile, m0, n0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    put(2 + round * 4);
    const int tn_idx = (int)(n0 / BNT);
    for (int kt = 0; kt < nk; kt += 2) {                       // nk is even (an odd count is padded with an all-zero K-tile)
      if constexpr (SUMA) duty = g.opsum && (kt & 3) == wc && (kt >> 2) % g.tiles_n == tn_idx;
      ktile(std::integral_constant<int, 0>{});
      if constexpr (SUMA) duty = g.opsum && ((kt + 1) & 3) == wc && ((kt + 1) >> 2) % g.tiles_n == tn_idx;
      ktile(std::integral_constant<int, 1>{});
    }
    if constexpr (SUMA) {
      // row t of the selector product = column sums of A fragment t: lanes 0..31 hold rows 0..7 (row = 4 (lane >> 4) + reg)
      if (lane < 32 && g.opsum) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t m = m0 + wr * 128 + (4 * (lane >> 4) + r) * 16 + (lane & 15);
          if (m < g.M && sacc[r] != 0.f) atomicAdd(g.opsum + m, sacc[r]);
        }
      }
      sacc = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    put(3 + round * 4);
    // ---- epilogue of this tile; the next tile's first pieces are already in flight
    float bcol[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int64_t col = n0 + wc * WN + j * 16 + (lane & 15);
      bcol[j] = (g.bias && col < g.N) ? g.bias[col] : 0.f;
    }
    epilogue<TC, NI, EPI>(g, acc, bcol, Cs, m0 + wr * 128, n0 + wc * WN, lane);
    put(4 + round * 4);
    // the stores above are younger than the pieces in flight: drain everything, so that the counted waits of the next
    // tile's first phases again count only pieces
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    put(5 + round * 4);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // pieces issued past the last tile (out of range: no data)
  if (wr == 0) __builtin_amdgcn_s_barrier();                   // balance the stagger
  if constexpr (STAMPS) { put(21); if (tid == 0 && stamps_buf) stamp[22] = __builtin_amdgcn_s_memrealtime(); }
}

template <typename TC, bool TA, bool TBK, int NI1 = 2, bool STAMPS = false, int ABL = 0, bool SUMA = false, int EPI = 0>
__global__ void __launch_bounds__(NT, 2) gemm256_kernel(Args args) {
  body<TC, TA, TBK, NI1, STAMPS, ABL, SUMA, false, EPI>(args.g, args.nk, blockIdx.x, gridDim.x, args.stamps);
}

// Several weight gradients in ONE launch: C_i[M_i, N_i] += A_i^T . B_i with A_i [K_i][M_i] (dY), B_i [K_i][N_i] (X), fp32 C,
// opsum_i (may be null) += column sums of A_i.  Every 256 x 192 tile of every problem is one workgroup that runs the whole
// reduction (no split-K, no atomics on C): a single gradient has 12-48 tiles, so the launch is a QUEUE of tiles - tiles
// tile0[i] .. of problem i, problems in the order given - and the caller cuts that queue into launches of 256 tiles (a
// gradient may straddle two launches: its tiles are independent).
// Placement: the hardware deals workgroups to the 8 XCDs round-robin (blockIdx & 7), and each XCD has its own L2.  Queue
// position q = (blockIdx & 7) * (grid / 8) + (blockIdx >> 3): every XCD takes a CONTIGUOUS run of the queue, and inside a
// problem the queue walks the shorter tile dimension fastest, so the ~32 tiles of an XCD form a compact rectangle that shares
// its operand panels in that L2 (8 x 4 tiles: 2.8 MB of operands per K-tile-row instead of 14 MB fetched tile by tile).
constexpr int MAX_GROUP = 12;
struct GroupArgs {
  GemmArgs gp[MAX_GROUP];
  int nk[MAX_GROUP];
  int first[MAX_GROUP + 1];   // queue position of each problem's first tile in this launch
  int tile0[MAX_GROUP];       // index (in the problem's own tile order) of that tile
  int nprob;
};
__global__ void __launch_bounds__(NT, 2) gemm256_group_kernel(GroupArgs ga) {
  const int per_xcd = (int)gridDim.x >> 3;
  const int q = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (q >= ga.first[ga.nprob]) return;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAX_GROUP; ++i)
    if (i < ga.nprob && q >= ga.first[i]) pi = i;
  const GemmArgs& g = ga.gp[pi];
  const int lt = q - ga.first[pi] + ga.tile0[pi];
  int tm, tn;
  if (g.tiles_m <= g.tiles_n) { tm = lt % g.tiles_m; tn = lt / g.tiles_m; }
  else { tn = lt % g.tiles_n; tm = lt / g.tiles_n; }
  body<float, true, false, 1, false, 0, true, true, 2>(g, ga.nk[pi], tm | (tn << 16), 8, nullptr);
}

}  // namespace t2

// The 256 x 256 kernel takes a problem when it is large enough to fill the chip with 256-wide tiles and its operands can be
// addressed by 32-bit byte offsets from a tile origin.  trans_a: A stored [K][M]; trans_b: B stored [N][K].
extern int g_opt_gemm256_min_tiles;                          // gemm.hip: neurst_set_option("gemm256_min_tiles", n)
extern int g_opt_gemm256_ablate;
extern int g_opt_gemm256_max_grid;
extern int g_opt_gemm256_bn;                                 // 0 = choose per problem, 192 / 256 = forced
static int t2_min_tiles() { return g_opt_gemm256_min_tiles; }
// Column width of the output tile: 256 or 192.  Cost model (measured, tools/exp/gemm256_driver): a round of 256-wide tiles
// costs its K loop plus a fixed ~15 % (epilogue, seam); a 192-wide tile 3/4 of that; the chip works in rounds of 256 tiles.
static int t2_pick_bn(const GemmArgs& g, int trans_a) {
  // k-strided A (weight gradients): the transposing LDS reads need more address registers, and the 256-wide body then
  // spills inside the K loop (a scratch reload drains the operand ring): 192 columns always
  if (trans_a) return 192;
  if (g_opt_gemm256_bn == 192 || g_opt_gemm256_bn == 256) return g_opt_gemm256_bn;
  const int64_t tm = cdiv64(g.M, t2::BM);
  double best = 0;
  int pick = 256;
  for (int bn : {256, 192}) {
    const int64_t tiles = tm * cdiv64(g.N, bn), rounds = cdiv64(tiles, 256);
    const double cost = (double)rounds * (bn / 256.0 + 0.15);
    if (best == 0 || cost < best - 1e-9) { best = cost; pick = bn; }
  }
  return pick;
}
bool neurst_gemm256_eligible(const GemmArgs& g, int trans_a, int trans_b, bool c_f32) {
  if (g.opsum && !(trans_a && !trans_b && c_f32)) return false;   // operand sums: weight gradients (A = dY [K][M]) only
  if (!g.a_aligned || !g.b_aligned || g.K % 8 || (trans_a && g.M % 8) || (!trans_b && g.N % 8)) return false;
  const int64_t tm = cdiv64(g.M, t2::BM), tn = cdiv64(g.N, t2_pick_bn(g, trans_a));
  if (tm * tn < t2_min_tiles()) return false;
  // Where it wins over the 128 x 128 kernels (tools/exp/gemm256_driver, M = 16000): an output tile costs ~10k cycles of
  // epilogue (all workgroups store at once: HBM-write bound) plus ~3300 cycles per K-tile, so long K loops pay - K = 3072
  // x1.13 (NT), K = 10000 x1.46 - and so does the fp32 vocabulary projection (x1.10); at K = 768 the 128-tile kernels,
  // whose co-resident workgroups hide each other's stores, stay ahead (x0.75-0.95).  min_tiles <= 1 (tests, drivers) lifts the rule.
  if (t2_min_tiles() > 1) {
    const bool long_k = !trans_a && ((g.K >= 2560 && trans_b) || g.K >= 8192);   // (weight gradients: grouped launches only)
    const bool wide_f32 = c_f32 && g.N >= 4096 && !trans_a;
    if (!long_k && !wide_f32) return false;
  }
  // 32-bit offsets inside one (tile, K-tile) window
  const int64_t wa = trans_a ? 64 * g.lda + 256 : 256 * g.lda + 64;
  const int64_t wb = trans_b ? 256 * g.ldb + 64 : 64 * g.ldb + 256;
  if (wa * 2 >= 0x7fff0000ll || wb * 2 >= 0x7fff0000ll) return false;
  return true;
}
static unsigned long long* g_stamps = nullptr;
// diagnostics (tools/exp/gemm256_driver): with a buffer of 256 x 24 uint64 set, NT launches with bf16 output run an
// instantiation that records clock readings per workgroup (entry, first compute, per tile: loop start / loop end /
// stores issued / stores drained, exit)
extern "C" void neurst_debug_gemm256_stamps(unsigned long long* buf) { g_stamps = buf; }
template <typename TC, int NI1>
static void launch256(const GemmArgs& g0, int trans_a, int trans_b, hipStream_t s) {
  t2::Args a;
  a.g = g0;
  a.g.tiles_m = (int)cdiv64(g0.M, t2::BM);
  a.g.tiles_n = (int)cdiv64(g0.N, 64 * (2 + NI1));
  a.nk = (int)((cdiv64(g0.K, t2::BK) + 1) / 2 * 2);       // even: the K loop runs two K-tiles per iteration
  a.stamps = g_stamps;
  a.ablate = g_opt_gemm256_ablate;
  const int ntiles = a.g.tiles_m * a.g.tiles_n;
  int grid = ntiles >= 256 ? 256 : (ntiles + 7) / 8 * 8;
  if (g_opt_gemm256_max_grid >= 8 && grid > g_opt_gemm256_max_grid) grid = g_opt_gemm256_max_grid / 8 * 8;   // experiment: leave CUs to another stream
  if constexpr (sizeof(TC) == 2) {
    if (g_stamps && !trans_a && trans_b) { t2::gemm256_kernel<TC, false, true, NI1, true><<<grid, t2::NT, 0, s>>>(a); return; }
    if constexpr (NI1 == 2) {
      if (g_opt_gemm256_ablate && !trans_a && trans_b) {
        switch (g_opt_gemm256_ablate) {
#define T2_ABL(m) case m: t2::gemm256_kernel<TC, false, true, NI1, false, m><<<grid, t2::NT, 0, s>>>(a); return;
          T2_ABL(1) T2_ABL(4) T2_ABL(8) T2_ABL(16) T2_ABL(64)
#undef T2_ABL
          default: break;
        }
      }
    }
  }
  if constexpr (sizeof(TC) == 4) {
    if (g0.opsum) { t2::gemm256_kernel<TC, true, false, NI1, false, 0, true><<<grid, t2::NT, 0, s>>>(a); return; }
  }
  const bool plain = !g0.relu_src && !g0.accumulate && g0.c_vec && g0.N % 8 == 0;      // EPI 1: see t2::epilogue
  if (!trans_a && trans_b && plain) t2::gemm256_kernel<TC, false, true, NI1, false, 0, false, 1><<<grid, t2::NT, 0, s>>>(a);
  else if (!trans_a && !trans_b && plain) t2::gemm256_kernel<TC, false, false, NI1, false, 0, false, 1><<<grid, t2::NT, 0, s>>>(a);
  else if (!trans_a && trans_b) t2::gemm256_kernel<TC, false, true, NI1><<<grid, t2::NT, 0, s>>>(a);
  else if (!trans_a && !trans_b) t2::gemm256_kernel<TC, false, false, NI1><<<grid, t2::NT, 0, s>>>(a);
  else if (trans_a && !trans_b) t2::gemm256_kernel<TC, true, false, NI1><<<grid, t2::NT, 0, s>>>(a);
  else t2::gemm256_kernel<TC, true, true, NI1><<<grid, t2::NT, 0, s>>>(a);
}
void neurst_gemm256_launch(const GemmArgs& g, int trans_a, int trans_b, int c_dtype, hipStream_t s) {
  const bool wide = t2_pick_bn(g, trans_a) == 256;
  if (c_dtype == NEURST_BF16) { if (wide) launch256<bf16_t, 2>(g, trans_a, trans_b, s); else launch256<bf16_t, 1>(g, trans_a, trans_b, s); }
  else { if (wide) launch256<float, 2>(g, trans_a, trans_b, s); else launch256<float, 1>(g, trans_a, trans_b, s); }
}

// Grouped weight gradients (see t2::gemm256_group_kernel).  Host arrays of n <= 12 problems; tiles tile_begin[i] ..
// tile_end[i] - 1 of problem i (null: all of them).  Returns false when a problem does not fit the kernel (alignment, 32-bit
// window) - the caller then falls back to one launch per gradient.
int neurst_gemm256_group_max() { return t2::MAX_GROUP; }
int64_t neurst_gemm256_group_tiles(int64_t M, int64_t N) { return cdiv64(M, t2::BM) * cdiv64(N, 192); }
// Per-problem predicate of the grouped kernel: alignment, whole 8-element chunks along both contiguous axes, 32-bit offsets
// inside one (tile, K-tile) window, tile coordinates that fit the packed block id.
bool neurst_gemm256_group_eligible(const GemmArgs& g) {
  if (!g.a_aligned || !g.b_aligned || g.M % 8 || g.N % 8) return false;
  if (!g.c_vec || !g.accumulate || g.relu_src || g.relu || g.bias) return false;   // its epilogue is C += in whole 16-byte chunks and nothing else
  if ((64 * g.lda + 256) * 2 >= 0x7fff0000ll || (64 * g.ldb + 256) * 2 >= 0x7fff0000ll) return false;
  if (cdiv64(g.M, t2::BM) > 0xffff || cdiv64(g.N, 192) > 0x7fff) return false;
  return true;
}
bool neurst_gemm256_group_launch(int n, const GemmArgs* probs, const int* tile_begin, const int* tile_end, hipStream_t s) {
  if (n < 1 || n > t2::MAX_GROUP) return false;
  t2::GroupArgs ga;
  memset(&ga, 0, sizeof(ga));
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const GemmArgs& g = probs[i];
    if (!neurst_gemm256_group_eligible(g)) return false;
    ga.gp[i] = g;
    ga.gp[i].tiles_m = (int)cdiv64(g.M, t2::BM);
    ga.gp[i].tiles_n = (int)cdiv64(g.N, 192);
    ga.gp[i].split_k = 1;
    ga.nk[i] = (int)((cdiv64(g.K, t2::BK) + 1) / 2 * 2);
    const int all = ga.gp[i].tiles_m * ga.gp[i].tiles_n;
    const int t0 = tile_begin ? tile_begin[i] : 0, t1 = tile_end ? tile_end[i] : all;
    if (t0 < 0 || t1 > all || t0 > t1) return false;
    ga.first[i] = total;
    ga.tile0[i] = t0;
    total += t1 - t0;
  }
  ga.first[n] = total;
  ga.nprob = n;
  if (total == 0) return true;
  t2::gemm256_group_kernel<<<(total + 7) / 8 * 8, t2::NT, 0, s>>>(ga);
  return true;
}
