// Fused attention backward for the encoder-size shapes (bf16, dh = 96 or 64, no causal mask): ONE kernel computes dQ, dK and
// dV with five MFMA products per (32-query slice, 256-key block) instead of the seven of the dq + dkv kernel pair of
// attention.hip, which stay the path for fp32 parity mode, causal masks, other head sizes and unaligned tensors.
// Reference arithmetic as there: st/models/tools/utils.py:545-553, st/layers/common_attention_v1.py:180-204,325-365.
//
// One workgroup = 4 waves (one per SIMD, the whole 512-register file) = one (batch, head).  It walks the keys in blocks of
// 256; wave w owns keys 64 w .. 64 w + 63 of the block (two 32-key MFMA columns) and keeps dK^T, dV^T of those keys in
// 192 accumulator registers while the workgroup sweeps the queries in slices of 32:
//   S  = Q.K^T, dP = dO.V^T   v_mfma_f32_32x32x16_bf16, key on the lane: A = rows of the Q / dO slice image (LDS, b128),
//                              B = K / V fragments held in registers for the whole block
//   P = exp2(c S - lse2), dS = P (keep/(1-rate) dP - delta)   in registers; the accumulator tiles are already the B operands
//                              of the next two products (k order permuted: query 16 t + 8 (j >> 2) + 4 g + (j & 3))
//   dV^T += dO^T.P, dK^T += Q^T.dS   A = transposed reads (ds_read_b64_tr_b16) of the same slice images
//   dQ  = dS.K                sums over the key = the lane index: dS crosses LDS once, as a [key][query] image read back
//                              transposed; K is a [key][d] LDS image of the block, read transposed as well
//                              (v_mfma_f32_16x16x32_bf16, wave w: query half w & 1, three (two) 16-wide d blocks)
// dQ needs the sum over key blocks: the workgroup owns all of them, so it carries the fp32 partial sums through a scratch
// tensor between its passes (same thread writes and reads an element; no atomics, deterministic); with Tk <= 256 there is
// one pass and no scratch traffic.  dQ of slice i - 1 is computed in iteration i, beside the vector work of slice i, so the
// loop needs ONE barrier per slice.  Q / dO slices and the K image arrive by LDS-DMA (buffer_load ... lds) with the
// bank swizzle applied on the source side; rows beyond the tensor read as zeros through the descriptor's range.
#include <algorithm>

#include "attention2.h"

#define NEG_BIAS (-1e9f)
// stops the scheduler from hoisting every LDS read of a phase to its start (registers); MFMA latency is covered by the chains' own length
#define FENCE() __builtin_amdgcn_sched_barrier(0)

namespace a2 {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct Args {
  const bf16_t *q, *k, *v, *o, *d_o;
  bf16_t *dq, *dk, *dv;
  const float* lse;
  float* delta;
  float* ws;                 // [ksplit][B*H][Tq][DH] fp32 partial dQ (only touched when Tk needs more than one key block)
  int64_t ws_part;           // floats per part of ws
  int ksplit;                // workgroups per (batch, head): part p takes key blocks p, p + ksplit, ...
  const uint32_t* keep;      // row-major keep bitmap (attention.hip), or null when rate == 0
  const int32_t* len_k;
  int64_t ldq, ldk, ldv, ldo, lddq, lddk, lddv;
  int64_t bsq, bsk, bsv, bso;
  int B, H, Tq, Tk, nw, keypad;
  float qscale, keep_scale;
  long long* stamps;         // diagnostic: s_memtime of workgroup 0 / wave 0 at phase boundaries (null in production)
};

__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned records) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, records, 0x00020000u};
}
// one LDS-DMA wave-instruction: 64 lanes x 16 B from rsrc + voff (per lane) to LDS bytes lds_addr .. + 1023, lane-linear
__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned lds_addr, int voff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}

// [rows][192 B] image (Q / dO slice, K block): 16-byte chunk ch of a row sits at chunk ch ^ ((row >> 2) & 3) - conflict-free
// for the b128 row reads of the 32x32x16 A operand and for the transposed reads of both MFMA shapes (tools/exp/lds_banks.py)
constexpr int ROWB = 192;
template <int RB = ROWB>
__device__ __forceinline__ int img_off(int row, int ch) { return row * RB + 16 * (ch ^ ((row >> 2) & 3)); }
// dS^T image [256 keys][32 queries] bf16, 64-byte rows, 8-byte slot s at slot s ^ ((row >> 1) & 7)
__device__ __forceinline__ int dst_off(int row, int slot) { return row * 64 + 8 * (slot ^ ((row >> 1) & 7)); }

__device__ __forceinline__ bf16x4 lds_tr(const char* p) {
  typedef bf16x4 __attribute__((address_space(3))) * lds_p;
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p));
}
__device__ __forceinline__ bf16x8 cat(bf16x4 lo, bf16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ bf16x8 pack8(const f32x16& x, int t) {
  return bf16x8{(bf16_t)x[8 * t], (bf16_t)x[8 * t + 1], (bf16_t)x[8 * t + 2], (bf16_t)x[8 * t + 3],
                (bf16_t)x[8 * t + 4], (bf16_t)x[8 * t + 5], (bf16_t)x[8 * t + 6], (bf16_t)x[8 * t + 7]};
}

// Workgroup -> (batch, head).  The hardware deals workgroups round-robin to the 8 XCDs (blockIdx & 7), each with its own L2; with
// head = blockIdx % H and H = 8 every XCD would run ONE head of every utterance.  A head's q / k / v rows are 192-byte (dh = 96)
// pieces of the packed [B T, 3 d] rows, i.e. 1.5 cache lines of 128 bytes: neighbouring heads share lines, and fetched from
// eight different L2s every shared line crosses the HBM side several times (measured by PMC: 200 MB per forward launch, 442 MB per
// backward launch against 115 / 215 MB algorithmic).  Each XCD therefore takes a CONTIGUOUS run of (batch, head) pairs: all heads of
// an utterance in one L2.
__device__ __forceinline__ int xcd_run(int bid, int n) {
  const int q = n >> 3, r = n & 7, xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

constexpr int KB = 256;        // keys per block
constexpr int QS = 32;         // queries per slice
constexpr int MAX_TQ = 2048;   // statistics of every query stay in LDS

template <int DH> struct Lds {
  static constexpr int CH = DH / 8;                       // 16-byte chunks per row
  static constexpr int ROWB = DH * 2;                     // row pitch of the images = the row itself (192 B at dh = 96, 128 B at 64)
  static constexpr int KIMG = 0, KIMG_B = KB * ROWB;
  static constexpr int QS0 = KIMG + KIMG_B, SLICE_B = QS * ROWB;
  static constexpr int DO0 = QS0 + 2 * SLICE_B;
  static constexpr int DST0 = DO0 + 2 * SLICE_B, DST_B = KB * 64;
  static constexpr int KW0 = DST0 + 2 * DST_B;            // [waves][key columns per wave][32 q] u32 = one word per key column of the block
  static constexpr int STAT0 = KW0 + (KB / 32) * 32 * 4;  // lse2 [Tqp], delta [Tqp]
  static constexpr int STG_ROWB = ROWB + 16;              // row pitch of the dK / dV store staging (208 / 144 B: the 8-byte column writes of
                                                          // a 192-B pitch land on two bank positions, 8-way)
  static_assert(KB * STG_ROWB <= DST0, "store staging overruns the K image + slice buffers");
  static int bytes(int tqp) { return STAT0 + 2 * tqp * 4; }
};

// NW = waves per workgroup: 4 (one per SIMD, 64 keys each) or 8 (two per SIMD, 32 keys each: one wave's exp / dropout /
// pack work runs under its partner's MFMA chains; 96 accumulator registers per wave instead of 192)
template <int DH, bool DROP, bool STAMPS, int NW>
__global__ void __launch_bounds__(NW * 64) mha2_bwd_kernel(Args a) {
  typedef Lds<DH> L;
  constexpr int ROWB = DH * 2;      // row pitch of the LDS images (the LDS-DMA writes rows back to back)
  constexpr int KS = DH / 16;       // k-steps of a product over d (32x32x16)
  constexpr int DB = DH / 32;       // 32-wide d blocks of dK^T / dV^T
  constexpr int NT = NW == 4 ? DH / 32 : 2;   // most 16-wide d blocks of dQ a wave computes (DH / 16 blocks x two query halves over the waves)
  constexpr int CH = DH / 8;
  constexpr int NTHR = NW * 64;
  constexpr int KPW = KB / NW;      // keys per wave: 64 or 32
  constexpr int NBLK = KPW / 32;    // 32-key MFMA columns per wave
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 31, g = lane >> 5;
  const int G = lane >> 4, qq = (lane & 15) >> 2, p4 = lane & 3;
  const int vid = xcd_run((int)blockIdx.x, (int)gridDim.x), bh = vid / a.ksplit, part = vid - bh * a.ksplit, b = bh / a.H, h = bh % a.H;
  const int Tq = a.Tq, Tk = a.Tk;
  const int nslice = (Tq + QS - 1) / QS, tqp = nslice * QS;
  float* lse2_s = reinterpret_cast<float*>(smem + L::STAT0);
  float* del_s = lse2_s + tqp;
  uint32_t* kw_s = reinterpret_cast<uint32_t*>(smem + L::KW0) + w * (32 * NBLK);
  typedef __attribute__((address_space(3))) char* lds_cp;
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_cp)smem;            // LDS byte address of the dynamic segment

  const bf16_t* qg = a.q + (int64_t)b * a.bsq + h * DH;
  const bf16_t* kg = a.k + (int64_t)b * a.bsk + h * DH;
  const bf16_t* vg = a.v + (int64_t)b * a.bsv + h * DH;
  const bf16_t* og = a.o + (int64_t)b * a.bso + h * DH;
  const bf16_t* dog = a.d_o + (int64_t)b * a.bso + h * DH;
  const float unit = 1.4426950408889634f, cs = a.qscale * unit;
  const int len = a.keypad ? a.len_k[b] : Tk;
  const int kv_end = (a.keypad && len > 0 && len < Tk) ? len : Tk;      // keys at and beyond it get exactly zero weight
  const int nkb = (kv_end + KB - 1) / KB;
  // Key blocks over several workgroups (fewer (batch, head) pairs than CUs: batch 16 at eight heads).  Part p sweeps the blocks
  // p, p + ksplit, ... and leaves its dQ as fp32 partial sums in its own slice of the workspace; mha2_dq_reduce_kernel, launched
  // behind this kernel, adds the slices up and writes dQ.  A part without a block (short utterance) has nothing to do.
  const int nparts = nkb < a.ksplit ? nkb : a.ksplit;                    // parts of this pair that own a key block
  if (part >= nparts) return;                                             // (workgroup-uniform, before any barrier or DMA)
  float* const wsq = a.ws + (int64_t)part * a.ws_part;
  int n_stamp = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    if constexpr (STAMPS) {
      FENCE();
      if (blockIdx.x == 0 && tid == 0) a.stamps[n_stamp++] = (long long)__builtin_amdgcn_s_memtime();
      FENCE();
    }
  };
  stamp();

  const u32x4 rs_q = make_rsrc(qg, (unsigned)(((int64_t)(Tq - 1) * a.ldq + DH) * 2));
  const u32x4 rs_do = make_rsrc(dog, (unsigned)(((int64_t)(Tq - 1) * a.ldo + DH) * 2));
  const u32x4 rs_k = make_rsrc(kg, (unsigned)(((int64_t)(Tk - 1) * a.ldk + DH) * 2));
  // slice image = QS * CH chunks; LDS slot n (16 B) = lane + 64 * (w + 4 * i) holds chunk (n % CH) ^ swizzle of row n / CH
  const int ldq2 = (int)a.ldq * 2, ldo2 = (int)a.ldo * 2;
  auto issue_slice = [&](int sl) __attribute__((always_inline)) {
    const int buf = sl & 1;
    int t = tid;
    asm volatile("" : "+v"(t));      // recomputed per call (a few VALU ops): as loop invariants these offsets get spilled, and a
                                     // scratch reload's vmcnt(0) then waits for the DMA issued just before it
#pragma unroll
    for (int i = 0; i < (QS * CH + NTHR - 1) / NTHR; ++i) {
      if (64 * (w + NW * i) < QS * CH) {                                  // wave-uniform
        const int n = t + NTHR * i, row = n / CH, ch = (n - row * CH) ^ ((row >> 2) & 3);
        const int qrow = sl * QS + row;
        const unsigned dst = (unsigned)(w + NW * i) * 1024u;
        dma16(rs_q, lds_base + L::QS0 + buf * L::SLICE_B + dst, qrow * ldq2 + ch * 16);
        dma16(rs_do, lds_base + L::DO0 + buf * L::SLICE_B + dst, qrow * ldo2 + ch * 16);
      }
    }
  };
  // keep words of slice sl for this wave's key columns: lane (c, g) fetches word g of query sl * 32 + c (zero beyond the tensor;
  // with one key column per wave the upper half-wave fetches nothing)
  auto load_kw = [&](int sl, int kb) -> uint32_t {
    uint32_t r = 0u;
    if constexpr (DROP) {
      const int q = sl * QS + c, wi = kb * (KB / 32) + NBLK * w + g;
      const bool ok = q < Tq && wi < a.nw && g < NBLK;
      const uint32_t* pw = a.keep + ((int64_t)bh * Tq + (ok ? q : 0)) * a.nw + (ok ? wi : 0);
      r = *pw;
      r = ok ? r : 0u;
    }
    return r;
  };

  // this wave's part of a dQ slice: 16 queries (half dq_qh) x ntw 16-wide d blocks from column dq_d0.  Four waves: DH / 32
  // blocks each.  Eight waves, DH = 96 (12 tiles): waves 0-3 take two blocks (columns 0..63), waves 4-7 one (64..95);
  // DH = 64 (8 tiles): one each.
  const int dq_qh = w & 1;
  const int ntw = NW == 4 ? NT : (DH == 96 && w < 4 ? 2 : 1);
  const int dq_d0 = NW == 4 ? (w >> 1) * (NT * 16) : (DH == 96 ? (w < 4 ? (w >> 1) * 32 : 64 + ((w - 4) >> 1) * 16) : (w >> 1) * 16);

  // ---- lane constants of the LDS addressing; every read below is one of these plus a compile-time offset
  // row reads (b128) of a [row][192 B] image, row = 32 m + c, chunk 2 ks + g = 4 (ks >> 1) + (2 (ks & 1) + g)
  const int sw_c = (c >> 2) & 3;
  const int rrd0 = c * ROWB + 16 * (g ^ sw_c), rrd1 = c * ROWB + 16 * ((2 + g) ^ sw_c);
  // transposed reads of a slice image for the 32x32x16 A operand: row 16 t + 4 g + qq (+ 8), chunk 4 db + 2 (G & 1) + (p4 >> 1)
  const int trs_lo = (4 * g + qq) * ROWB + 16 * ((2 * (G & 1) + (p4 >> 1)) ^ g) + 8 * (p4 & 1);
  const int trs_hi = (4 * g + qq + 8) * ROWB + 16 * ((2 * (G & 1) + (p4 >> 1)) ^ (g + 2)) + 8 * (p4 & 1);
  // transposed reads of the dS^T image for the 16x16x32 A operand: row 32 kk + 8 G + qq (+ 4), slot 4 qh + p4
  const int trd_lo = dst_off(8 * G + qq, 4 * dq_qh + p4), trd_hi = dst_off(8 * G + qq + 4, 4 * dq_qh + p4);
  // transposed reads of the K image for the 16x16x32 B operand: row 32 kk + 8 G + qq (+ 4), chunk dq_d0 / 8 + 2 n + (p4 >> 1)
  int trk_lo[NT], trk_hi[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    // (a wave with fewer than NT blocks of its own multiplies its last block once more - uniform code, the result is dropped)
    const int nb = n < ntw ? n : ntw - 1;
    trk_lo[n] = img_off<ROWB>(8 * G + qq, dq_d0 / 8 + 2 * nb + (p4 >> 1)) + 8 * (p4 & 1);
    trk_hi[n] = img_off<ROWB>(8 * G + qq + 4, dq_d0 / 8 + 2 * nb + (p4 >> 1)) + 8 * (p4 & 1);
  }
  // dS^T stores: key row KPW w + 32 blk + c, slot 2 j + g
  const int krow = KPW * w + c;

  // K image of a key block by LDS-DMA (offsets from an opaque copy of the thread index: kept as loop invariants they are
  // spilled, and each reload's vmcnt(0) waits for the DMA piece issued just before it - the pieces would go out one by one)
  auto issue_kimage = [&](int kb) __attribute__((always_inline)) {
    int t = tid;
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int i = 0; i < KB * CH / NTHR; ++i) {
      const int n = t + NTHR * i, row = n / CH, ch = (n - row * CH) ^ ((row >> 2) & 3);
      dma16(rs_k, lds_base + L::KIMG + (unsigned)(w + NW * i) * 1024u, (int)((kb * KB + row) * a.ldk * 2 + ch * 16));
    }
  };
  // the first block's K image and query slice are on their way while the statistics below are computed
  issue_kimage(part);
  issue_slice(0);

  // ---- statistics of every query: lse2 = lse * log2(e) (+inf beyond Tq: p = 0), delta = rowsum(dO * O).  Four lanes per
  // row (chunks p4, p4 + 4, ...: 64 contiguous bytes per row and instruction), quad sum by DPP.  (Streaming them one
  // slice ahead inside the first pass instead was measured: the burst only moves into the loop, no gain.)
  // Four row groups per iteration, every load issued before the first sum: one memory latency per iteration instead of four.
  {
    constexpr int RPI = NTHR / 4;                                          // rows per group
    for (int q0 = 0; q0 < tqp; q0 += 4 * RPI) {
      bf16x8 x[4][CH / 4], y[4][CH / 4];
      float l2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * RPI + (tid >> 2);
        const int qc = q < Tq ? q : Tq - 1;
        const bf16x8* po = reinterpret_cast<const bf16x8*>(og + (int64_t)qc * a.ldo);
        const bf16x8* pd = reinterpret_cast<const bf16x8*>(dog + (int64_t)qc * a.ldo);
#pragma unroll
        for (int i = 0; i < CH / 4; ++i) { x[u][i] = po[p4 + 4 * i]; y[u][i] = pd[p4 + 4 * i]; }
        l2[u] = a.lse[(int64_t)bh * Tq + qc];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * RPI + (tid >> 2);
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < CH / 4; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) d += (float)x[u][i][j] * (float)y[u][i][j];
        d += NEURST_DPP_F32(d, 0xB1, 0xF);    // quad_perm [1,0,3,2]
        d += NEURST_DPP_F32(d, 0x4E, 0xF);    // quad_perm [2,3,0,1]
        if (p4 == 0 && q < tqp) {
          if (q < Tq) a.delta[(int64_t)bh * Tq + q] = d;
          lse2_s[q] = q < Tq ? l2[u] * unit : INFINITY;
          del_s[q] = q < Tq ? d : 0.f;
        }
      }
    }
  }

  for (int kb = part; kb < nkb; kb += a.ksplit) {
    const int K0 = kb * KB;
    if (kb != part) {
      __syncthreads();                                                    // previous pass has left the K image and the staging buffers
      issue_kimage(kb);
      issue_slice(0);
    }
    stamp();
    int tk = tid;
    asm volatile("" : "+v"(tk));
    // ---- V fragments of this wave's keys (B operands: column = key, k = 16 ks + 8 g + j); K fragments come from the image
    bf16x8 vf[NBLK][KS];
    float kbias[NBLK];
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      const int key = K0 + KPW * w + 32 * blk + (tk & 31);
      const bool ok = key < Tk;
      const int kc = ok ? key : Tk - 1;
      const bf16x8 zero = {};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 vv = *reinterpret_cast<const bf16x8*>(vg + (int64_t)kc * a.ldv + 16 * ks + 8 * ((tk >> 5) & 1));
        vf[blk][ks] = ok ? vv : zero;
      }
      kbias[blk] = !ok ? -INFINITY : ((a.keypad && key >= len) ? NEG_BIAS * unit : 0.f);
    }
    f32x16 dk[NBLK][DB], dv[NBLK][DB];
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int v = 0; v < 16; ++v) { dk[blk][db][v] = 0.f; dv[blk][db][v] = 0.f; }
    uint32_t kw_next = load_kw(0, kb);
    const bool first = kb == part, last = kb + a.ksplit >= nkb && nparts == 1;   // last: dQ is complete in this pass

    // dQ of slice sl from dsT[sl & 1] and the K image, in KB / 32 k-steps that the slice loop places between the groups of
    // its vector work; added to / stored over the partial sums (workspace rows padded to whole slices)
    f32x4 dqa[NT], dqw[NT];
    bf16x8 dq_af[2], dq_bf[2][NT];
    auto dq_fetch = [&](int sl, int kk) __attribute__((always_inline)) {
      const char* dst = smem + L::DST0 + (sl & 1) * L::DST_B;
      const char* kimq = smem + L::KIMG;
      dq_af[kk & 1] = cat(lds_tr(dst + trd_lo + kk * 32 * 64), lds_tr(dst + trd_hi + kk * 32 * 64));
#pragma unroll
      for (int n = 0; n < NT; ++n) dq_bf[kk & 1][n] = cat(lds_tr(kimq + trk_lo[n] + kk * 32 * ROWB), lds_tr(kimq + trk_hi[n] + kk * 32 * ROWB));
    };
    // the partial sums of the earlier passes: loaded FIRST in an iteration (older than the DMA pieces issued after them, so
    // the wait for them leaves those in flight) and added after the last k-step
    // (addresses from an opaque copy of the lane index: as loop invariants the compiler keeps a 64-bit pointer per store in
    // scratch, and every reload's s_waitcnt vmcnt(0) then waits for the store issued just before it - one HBM round trip
    // per store instruction instead of one per group)
    auto dq_load_partial = [&](int sl) __attribute__((always_inline)) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int q_base = sl * QS + 16 * dq_qh + 4 * (ln >> 4);           // + r: the four queries of this lane's accumulator registers
      const float* wsp = wsq + ((int64_t)bh * tqp + q_base) * DH + dq_d0 + (ln & 15);
#pragma unroll
      for (int n = 0; n < NT; ++n)
        if (n < ntw) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dqw[n][r] = wsp[r * DH + 16 * n];
        }
    };
    auto dq_begin = [&](int sl) __attribute__((always_inline)) {
#pragma unroll
      for (int n = 0; n < NT; ++n) dqa[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      dq_fetch(sl, 0);
    };
    auto dq_step = [&](int sl, int kk) __attribute__((always_inline)) {
      if (kk + 1 < KB / 32) dq_fetch(sl, kk + 1);
#pragma unroll
      for (int n = 0; n < NT; ++n) dqa[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dq_af[kk & 1], dq_bf[kk & 1][n], dqa[n], 0, 0, 0);
    };
    auto dq_end = [&](int sl) __attribute__((always_inline)) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int q_base = sl * QS + 16 * dq_qh + 4 * (ln >> 4);
      if (!first) {
#pragma unroll
        for (int n = 0; n < NT; ++n) dqa[n] += dqw[n];
      }
      if (last) {
        bf16_t* dqp = a.dq + ((int64_t)b * Tq + q_base) * a.lddq + h * DH + dq_d0 + (ln & 15);
        if (sl * QS + QS <= Tq) {
#pragma unroll
          for (int n = 0; n < NT; ++n)
            if (n < ntw) {
#pragma unroll
              for (int r = 0; r < 4; ++r) dqp[r * a.lddq + 16 * n] = (bf16_t)(dqa[n][r] * a.qscale);
            }
        } else {
#pragma unroll
          for (int n = 0; n < NT; ++n)
            if (n < ntw) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (q_base + r < Tq) dqp[r * a.lddq + 16 * n] = (bf16_t)(dqa[n][r] * a.qscale);
            }
        }
      } else {
        float* wsp = wsq + ((int64_t)bh * tqp + q_base) * DH + dq_d0 + (ln & 15);
#pragma unroll
        for (int n = 0; n < NT; ++n)
          if (n < ntw) {
#pragma unroll
            for (int r = 0; r < 4; ++r) wsp[r * DH + 16 * n] = dqa[n][r];
          }
      }
    };

    for (int sl = 0; sl < nslice; ++sl) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // this wave's DMA pieces of slice sl (and of the K image) have landed
      __syncthreads();
      if (sl < 6) stamp();
      const int slq = sl > 0 ? sl - 1 : 0;
      if (!first) dq_load_partial(slq);
      if (sl + 1 < nslice) issue_slice(sl + 1);
      // keep words: this wave's [2 blk][32 q] table, written by the lanes that fetched them one slice ahead
      if constexpr (DROP) {
        if (NBLK == 2 || lane < 32) kw_s[lane] = kw_next;
        kw_next = load_kw(sl + 1, kb);
      }
      const char* qs = smem + L::QS0 + (sl & 1) * L::SLICE_B;
      const char* dos = smem + L::DO0 + (sl & 1) * L::SLICE_B;
      const char* kim = smem + L::KIMG + (KPW * w) * ROWB;
      // dQ of the previous slice runs as filler: its k-steps sit between the groups of vector work below (at sl == 0 it
      // multiplies whatever the buffer holds and the result is dropped)
      dq_begin(slq);
      if (sl == 4) stamp();
      // ---- per key column: S = Q.K^T, dP = dO.V^T, then P and dS in registers and dS^T into LDS for the next iteration's dQ
      bf16x8 pf[NBLK][2], dsf[NBLK][2];
      char* dstw = smem + L::DST0 + (sl & 1) * L::DST_B;
#pragma unroll
      for (int blk = 0; blk < NBLK; ++blk) {
        f32x16 s, dp;
#pragma unroll
        for (int v = 0; v < 16; ++v) { s[v] = 0.f; dp[v] = 0.f; }
        {
          auto ro_of = [&](int ks) __attribute__((always_inline)) { return ((ks & 1) ? rrd1 : rrd0) + 64 * (ks >> 1); };
          const char* kimb = kim + 32 * blk * ROWB;
          // fragments two k-steps ahead: a step's two MFMAs (64 cycles) are shorter than an LDS round trip
          // (one wave per SIMD: two steps ahead; two per SIMD: one step ahead, the partner covers the rest - 12 registers)
          constexpr int FD = NW == 4 ? 3 : 2;
          bf16x8 aq[FD], ad[FD], kf[FD];
          auto fetch = [&](int ks) __attribute__((always_inline)) {
            const int ro = ro_of(ks);
            aq[ks % FD] = *reinterpret_cast<const bf16x8*>(qs + ro);
            ad[ks % FD] = *reinterpret_cast<const bf16x8*>(dos + ro);
            kf[ks % FD] = *reinterpret_cast<const bf16x8*>(kimb + ro);
          };
#pragma unroll
          for (int ks = 0; ks < FD - 1; ++ks) fetch(ks);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            if (ks + FD - 1 < KS) fetch(ks + FD - 1);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks % FD], kf[ks % FD], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad[ks % FD], vf[blk][ks], dp, 0, 0, 0);
            FENCE();
          }
        }
        FENCE();
        if (sl == 4 && blk == 0) stamp();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse2_s + sl * QS + 8 * j + 4 * g);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + sl * QS + 8 * j + 4 * g);
          u32x4 k4 = {0u, 0u, 0u, 0u};
          if constexpr (DROP) k4 = *reinterpret_cast<const u32x4*>(kw_s + 32 * blk + 8 * j + 4 * g);
          // the KB / 32 k-steps of dQ, spread over the NBLK x 4 groups of vector work
#pragma unroll
          for (int u = 0; u < 2 / NBLK; ++u) dq_step(slq, (2 / NBLK) * (4 * blk + j) + u);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int v = 4 * j + r;
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[v], cs, kbias[blk] - l4[r]));
            float pd = p;
            if constexpr (DROP) {
              const int m = __builtin_amdgcn_sbfe(k4[r], c, 1);            // 0 or -1 from bit c
              pd = __builtin_bit_cast(float, __builtin_bit_cast(int, p) & m);
            }
            const float t = p * d4[r];
            s[v] = pd;
            dp[v] = DROP ? __builtin_fmaf(pd * dp[v], a.keep_scale, -t) : __builtin_fmaf(p, dp[v], -t);
          }
          FENCE();
        }
        if (sl == 4 && blk == 0) stamp();
#pragma unroll
        for (int t = 0; t < 2; ++t) { pf[blk][t] = pack8(s, t); dsf[blk][t] = pack8(dp, t); }
        // key row 64 w + 32 blk + c, queries 8 j + 4 g .. + 3 = slot 2 j + g
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bf16x4 d4 = {dsf[blk][j >> 1][4 * (j & 1)], dsf[blk][j >> 1][4 * (j & 1) + 1], dsf[blk][j >> 1][4 * (j & 1) + 2],
                             dsf[blk][j >> 1][4 * (j & 1) + 3]};
          *reinterpret_cast<bf16x4*>(dstw + dst_off(krow + 32 * blk, 2 * j + g)) = d4;
        }
        FENCE();
      }
      if (sl > 0) dq_end(sl - 1);
      if (sl == 4) stamp();
      // ---- dV^T += dO^T.P, dK^T += Q^T.dS: A = transposed reads, rows 16 t + 4 g + q (+ 8), columns 32 db + 16 (G & 1) + 4 p
      {
        auto o2_of = [&](int i) __attribute__((always_inline)) { return 16 * (i / DB) * ROWB + 64 * (i % DB); };
        bf16x8 adt = cat(lds_tr(dos + trs_lo + o2_of(0)), lds_tr(dos + trs_hi + o2_of(0)));
        bf16x8 aqt = cat(lds_tr(qs + trs_lo + o2_of(0)), lds_tr(qs + trs_hi + o2_of(0)));
#pragma unroll
        for (int i = 0; i < 2 * DB; ++i) {
          const int t = i / DB, db = i % DB;
          bf16x8 nadt = adt, naqt = aqt;
          if (i + 1 < 2 * DB) {
            nadt = cat(lds_tr(dos + trs_lo + o2_of(i + 1)), lds_tr(dos + trs_hi + o2_of(i + 1)));
            naqt = cat(lds_tr(qs + trs_lo + o2_of(i + 1)), lds_tr(qs + trs_hi + o2_of(i + 1)));
          }
#pragma unroll
          for (int blk = 0; blk < NBLK; ++blk) {
            dv[blk][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(adt, pf[blk][t], dv[blk][db], 0, 0, 0);
            dk[blk][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aqt, dsf[blk][t], dk[blk][db], 0, 0, 0);
          }
          FENCE();
          adt = nadt; aqt = naqt;
        }
      }
    }
    stamp();
    __syncthreads();
    if (!first) dq_load_partial(nslice - 1);
    dq_begin(nslice - 1);
#pragma unroll
    for (int kk = 0; kk < KB / 32; ++kk) { dq_step(nslice - 1, kk); FENCE(); }
    dq_end(nslice - 1);
    __syncthreads();
    stamp();
    if (false) __syncthreads();                                                    // every wave is done with the K image: it becomes the store staging area
    // ---- dK, dV of this wave's keys: registers -> [KPW][DH] bf16 tile in LDS (padded rows) -> whole rows to memory
    {
      constexpr int SRB = L::STG_ROWB;
      char* st = smem + L::KIMG + w * (KPW * SRB);
      int ln = lane;                                                      // opaque: see dq_load_partial
      asm volatile("" : "+v"(ln));
      const int lc = ln & 31, lgh = ln >> 5;
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        const float sc = which == 0 ? a.qscale : (DROP ? a.keep_scale : 1.f);
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
          for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f32x16& x = which == 0 ? dk[blk][db] : dv[blk][db];
              const bf16x4 o4 = {(bf16_t)(x[4 * j] * sc), (bf16_t)(x[4 * j + 1] * sc), (bf16_t)(x[4 * j + 2] * sc), (bf16_t)(x[4 * j + 3] * sc)};
              *reinterpret_cast<bf16x4*>(st + (32 * blk + lc) * SRB + (32 * db + 8 * j + 4 * lgh) * 2) = o4;
            }
        const int64_t ldo = which == 0 ? a.lddk : a.lddv;
        bf16_t* outp = (which == 0 ? a.dk : a.dv) + ((int64_t)b * Tk + K0 + KPW * w) * ldo + h * DH;   // wave-uniform part
        const int krem = Tk - (K0 + KPW * w);                             // rows of this wave's tile inside the tensor
        bf16x8 val[KPW * CH / 64];
#pragma unroll
        for (int i = 0; i < KPW * CH / 64; ++i) {
          const int n = ln + 64 * i, row = n / CH, ch = n - row * CH;
          val[i] = *reinterpret_cast<const bf16x8*>(st + row * SRB + ch * 16);
        }
#pragma unroll
        for (int i = 0; i < KPW * CH / 64; ++i) {
          const int n = ln + 64 * i, row = n / CH, ch = n - row * CH;
          if (row < krem) *reinterpret_cast<bf16x8*>(outp + (int64_t)row * ldo + ch * 8) = val[i];
        }
      }
    }
  }
  stamp();
  // ---- keys the passes did not visit (behind the valid length): exactly zero gradients
  if (part == 0) {
    const bf16x8 zero = {};
    const int k_first = nkb * KB;
    for (int n = tid; n < (Tk - k_first) * CH; n += NTHR) {
      const int key = k_first + n / CH, ch = n % CH;
      *reinterpret_cast<bf16x8*>(a.dk + ((int64_t)b * Tk + key) * a.lddk + h * DH + ch * 8) = zero;
      *reinterpret_cast<bf16x8*>(a.dv + ((int64_t)b * Tk + key) * a.lddv + h * DH + ch * 8) = zero;
    }
  }
}

// dQ of pairs whose key blocks went to several workgroups (a.ksplit > 1): bf16(qscale x the sum of the parts' fp32 slices), as a
// launch of its own behind the backward kernel.  (First version: the part of a pair that finished last - an arrival counter per
// pair behind __threadfence() - summed them inside the kernel.  Measured slower than no split at all, batch 16: 8.78 -> 8.83 ms
// per step against 8.64 with the forward split alone; a device-scope release on this chip writes the XCD's L2 back, from every
// wave of every workgroup, and one workgroup then walks 384 KB alone.  A kernel boundary orders the same data for free and
// puts the whole chip on the sum.)
template <int DH>
__global__ void __launch_bounds__(256) mha2_dq_reduce_kernel(Args a) {
  constexpr int C4 = DH / 4;
  const int tqp = (a.Tq + QS - 1) / QS * QS;
  const int64_t total = (int64_t)a.B * a.H * a.Tq * C4;
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < total; n += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(n % C4);
    const int64_t row = n / C4;                                           // (pair, query)
    const int q = (int)(row % a.Tq), bh = (int)(row / a.Tq), b = bh / a.H, h = bh - b * a.H;
    const int len = a.keypad ? a.len_k[b] : a.Tk;
    const int kv_end = (a.keypad && len > 0 && len < a.Tk) ? len : a.Tk;
    const int nkb = (kv_end + KB - 1) / KB, nparts = nkb < a.ksplit ? nkb : a.ksplit;
    if (nparts == 1) continue;                                            // one part: its kernel wrote dQ itself
    const float* src = a.ws + ((int64_t)bh * tqp + q) * DH + 4 * c4;
    f32x4 sum = *reinterpret_cast<const f32x4*>(src);
    for (int pp = 1; pp < nparts; ++pp) sum += *reinterpret_cast<const f32x4*>(src + (int64_t)pp * a.ws_part);
    const bf16x4 o4 = {(bf16_t)(sum[0] * a.qscale), (bf16_t)(sum[1] * a.qscale), (bf16_t)(sum[2] * a.qscale), (bf16_t)(sum[3] * a.qscale)};
    *reinterpret_cast<bf16x4*>(a.dq + ((int64_t)b * a.Tq + q) * a.lddq + h * DH + 4 * c4) = o4;
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Forward for the same shapes (bf16, dh = 96, no causal mask), "query on the lane":
//   S^T = K.Q^T   v_mfma_f32_32x32x16_bf16: A = rows of a 32-key tile of the K image (LDS, b128), B = this wave's 32 queries,
//                 held in registers for the whole sweep.  A lane owns ONE query (column c) and 16 of the tile's 32 keys:
//                 running maximum and sum are per-lane scalars, one exchange between the two 32-lane halves per tile.
//   P^T -> bf16   the accumulator tile is already the B operand of the second product (keys in the permuted k order
//                 16 t + 8 (i >> 2) + 4 g + (i & 3), as in the backward above)
//   O^T += V^T.P^T   A = transposed reads (ds_read_b64_tr_b16) of the V image; the rescale by exp2(m_old - m_new) is a
//                 per-lane factor on the accumulator columns and is skipped while no lane's maximum moved.
// One workgroup = 8 waves (two per SIMD: one wave's vector work - exp2, dropout words - runs under the other's MFMAs) = one
// (batch, head); a wave sweeps all keys for one 32-query slice, then takes slice + 8.  K / V arrive in blocks of 128 keys by
// LDS-DMA, double-buffered, one barrier per block.  Dropout decisions are drawn 32 at a time (common.h:attn_keep_word): the
// two halves of a wave draw the words of two consecutive key tiles and swap them (v_permlane32_swap); every drawn word is
// also the bitmap word the backward reads.
// x of the lower 32 lanes -> lo (in every lane), x of the upper 32 lanes -> hi: v_permlane32_swap swaps lanes 32..63 of its
// first operand with lanes 0..31 of its second.  The second operand is made opaque: given the same value twice the
// compiler folds the two results into one (observed: "v_add v2, v139, v139" for sw[0] + sw[1]).
__device__ __forceinline__ void halves(unsigned x, unsigned& lo, unsigned& hi) {
  unsigned y = x;
  asm volatile("" : "+v"(y));
  const auto sw = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  lo = sw[0];
  hi = sw[1];
}
__device__ __forceinline__ void halves(float x, float& lo, float& hi) {
  unsigned a, b;
  halves(__builtin_bit_cast(unsigned, x), a, b);
  lo = __builtin_bit_cast(float, a);
  hi = __builtin_bit_cast(float, b);
}
constexpr int KBF = 128;       // keys per block
constexpr int FW = 8;          // waves
template <int DH> struct LdsF {
  static constexpr int ROWB = DH * 2;
  static constexpr int IMG_B = KBF * ROWB;
  static constexpr int K0 = 0, V0 = 2 * IMG_B;              // [2 buffers] each
  static constexpr int OST0 = 4 * IMG_B, OST_B = QS * ROWB;  // per-wave output staging tile
  static constexpr int BYTES = OST0 + FW * OST_B;
};
struct FwdArgs {
  const bf16_t *q, *k, *v;
  bf16_t* o;
  float* lse;
  uint32_t* keep;            // bitmap to fill (rate > 0), or null
  int keep_ready;            // the bitmap already holds the decisions (neurst_mha_keep_bits): load instead of drawing
  const int32_t* len_k;
  int64_t ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso;
  int B, H, Tq, Tk, nw, keypad, idx32;
  int qsplit;                // workgroups per (batch, head): workgroup part p takes the groups of eight query slices p, p + qsplit, ...
  uint32_t key, inner0, t16;
  float qscale, keep_scale;
};

// READY: the keep bitmap already holds the decisions (a.keep_ready, checked by the host): they are loaded, not drawn.  A template
// argument, not a run-time flag: the loop body then holds one of the two paths instead of both.
template <int DH, bool DROP, int ABL = 0, bool READY = false>
__global__ void __launch_bounds__(512, 1) mha2_fwd_kernel(FwdArgs a) {
  typedef LdsF<DH> L;
  constexpr int ROWB = DH * 2;
  constexpr int KS = DH / 16, DB = DH / 32, CH = DH / 8;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 31, g = lane >> 5;
  const int G = lane >> 4, qq = (lane & 15) >> 2, p4 = lane & 3;
  // (the parts of one (batch, head) are neighbours in the XCD's run: they read the same K / V rows through one L2)
  const int vid = xcd_run((int)blockIdx.x, (int)gridDim.x), bh = vid / a.qsplit, part = vid - bh * a.qsplit, b = bh / a.H, h = bh % a.H;
  const int Tq = a.Tq, Tk = a.Tk;
  typedef __attribute__((address_space(3))) char* lds_cp;
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_cp)smem;
  const bf16_t* qg = a.q + (int64_t)b * a.bsq + h * DH;
  const bf16_t* kg = a.k + (int64_t)b * a.bsk + h * DH;
  const bf16_t* vg = a.v + (int64_t)b * a.bsv + h * DH;
  const float unit = 1.4426950408889634f, cs = a.qscale * unit;
  const int len = a.keypad ? a.len_k[b] : Tk;
  const int kv_end = (a.keypad && len > 0 && len < Tk) ? len : Tk;       // keys at and beyond it get exactly zero weight
  const int kv_lim = a.keypad ? (len < Tk ? len : Tk) : Tk;              // keys at and beyond it carry a bias
  const int nkb = (kv_end + KBF - 1) / KBF;
  const int nslice = (Tq + QS - 1) / QS, ngroup = (nslice + FW - 1) / FW;
  const u32x4 rs_k = make_rsrc(kg, (unsigned)(((int64_t)(Tk - 1) * a.ldk + DH) * 2));
  const u32x4 rs_v = make_rsrc(vg, (unsigned)(((int64_t)(Tk - 1) * a.ldv + DH) * 2));
  const int ldk2 = (int)a.ldk * 2, ldv2 = (int)a.ldv * 2;
  // block image = KBF * CH chunks of 16 B; LDS slot n = tid + 512 i holds chunk (n % CH) ^ swizzle of row n / CH
  auto issue_block = [&](int kb, int buf) __attribute__((always_inline)) {
    int t = tid;
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int i = 0; i < KBF * CH / 512; ++i) {
      const int n = t + 512 * i, row = n / CH, ch = (n - row * CH) ^ ((row >> 2) & 3);
      const unsigned dst = (unsigned)(w + FW * i) * 1024u + (unsigned)buf * L::IMG_B;
      dma16(rs_k, lds_base + L::K0 + dst, (kb * KBF + row) * ldk2 + ch * 16);
      dma16(rs_v, lds_base + L::V0 + dst, (kb * KBF + row) * ldv2 + ch * 16);
    }
  };
  // lane constants of the LDS addressing (as in the backward: row reads for the 32x32x16 A operand, transposed reads)
  const int sw_c = (c >> 2) & 3;
  const int rrd0 = c * ROWB + 16 * (g ^ sw_c), rrd1 = c * ROWB + 16 * ((2 + g) ^ sw_c);
  const int trs_lo = (4 * g + qq) * ROWB + 16 * ((2 * (G & 1) + (p4 >> 1)) ^ g) + 8 * (p4 & 1);
  const int trs_hi = (4 * g + qq + 8) * ROWB + 16 * ((2 * (G & 1) + (p4 >> 1)) ^ (g + 2)) + 8 * (p4 & 1);
  char* const ost = smem + L::OST0 + w * L::OST_B;

  // Stored dropout decisions (keep_ready): the words of a block are requested one block AHEAD, before that block's LDS-DMA
  // pieces are issued.  The vector-memory counter retires in order and hipcc does not count the DMA statements: a keep word
  // loaded behind them (as it was, inside the tile loop) made its "vmcnt(0)" wait for the whole next K / V block as well.
  uint32_t kwn[2] = {0u, 0u};
  auto load_kw = [&](int grp_, int kb_) __attribute__((always_inline)) {
    const int q_ = (grp_ * FW + w) * QS + c;
    const bool ok = q_ < Tq;
    const int64_t rg = (int64_t)bh * Tq + (ok ? q_ : 0);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int wi = kb_ * (KBF / 32) + 2 * pr + g;
      kwn[pr] = (ok && wi < a.nw) ? a.keep[rg * a.nw + wi] : 0u;
    }
  };
  if constexpr (DROP && READY) load_kw(part, 0);
  issue_block(0, 0);
  int it = 0;                                                             // blocks consumed so far (buffer = it & 1)
  for (int grp = part; grp < ngroup; grp += a.qsplit) {
    const int sl = grp * FW + w;                                          // this wave's slice (may lie beyond the last one: computes on zeros)
    const int q = sl * QS + c;
    const bool q_ok = q < Tq;
    const int64_t row_g = (int64_t)bh * Tq + (q_ok ? q : 0);
    bf16x8 qf[KS];
    {
      const bf16x8 zero = {};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(qg + (int64_t)(q_ok ? q : 0) * a.ldq + 16 * ks + 8 * g);
        qf[ks] = q_ok ? x : zero;
      }
    }
    f32x16 oacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int v = 0; v < 16; ++v) oacc[db][v] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int kb = 0; kb < nkb; ++kb, ++it) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // this wave's pieces of block `it` have landed
      __syncthreads();                                                    // everyone's have, and everyone has left the other buffer
      uint32_t kwc0 = kwn[0], kwc1 = kwn[1];                              // this block's stored keep words (landed: waited for above)
      {
        const int nkb_next = kb + 1 < nkb ? kb + 1 : 0;
        const bool more = kb + 1 < nkb || grp + a.qsplit < ngroup;
        if constexpr (DROP) {
          asm volatile("" : "+v"(kwc0), "+v"(kwc1));                      // their use stays in front of the next loads and DMA pieces
          if (READY && more) load_kw(kb + 1 < nkb ? grp : grp + a.qsplit, nkb_next);
        }
        if (more) issue_block(nkb_next, (it + 1) & 1);
      }
      const char* kim = smem + L::K0 + (it & 1) * L::IMG_B;
      const char* vim = smem + L::V0 + (it & 1) * L::IMG_B;
      const int K0 = kb * KBF;
      // S^T of tile kt, one k-step (MFMA) at a time.  (Issuing the product of tile kt + 1 in pieces between the groups of the
      // vector work of tile kt was measured: 45 -> 50 us at T = 500 - the second wave on the SIMD already fills the matrix
      // pipe, the extra uniform branches and scheduling fences only cost.)
      f32x16 sn;
      bf16x8 kf[3];
      auto ro_of = [&](int ks) __attribute__((always_inline)) { return ((ks & 1) ? rrd1 : rrd0) + 64 * (ks >> 1); };
      auto s_begin = [&](int kt) __attribute__((always_inline)) {
        const char* kimb = kim + 32 * kt * ROWB;
#pragma unroll
        for (int v = 0; v < 16; ++v) sn[v] = 0.f;
        kf[0] = *reinterpret_cast<const bf16x8*>(kimb + ro_of(0));
        kf[1] = *reinterpret_cast<const bf16x8*>(kimb + ro_of(1));
      };
      auto s_step = [&](int kt, int ks) __attribute__((always_inline)) {
        const char* kimb = kim + 32 * kt * ROWB;
        if (ks + 2 < KS) kf[(ks + 2) % 3] = *reinterpret_cast<const bf16x8*>(kimb + ro_of(ks + 2));
        if constexpr ((ABL & 4) == 0) sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks % 3], qf[ks], sn, 0, 0, 0);
        else asm volatile("" ::"v"(kf[ks % 3]));
      };
#pragma unroll 1
      for (int pr = 0; pr < 2; ++pr) {
        if (K0 + 64 * pr >= kv_end) break;                                // wave-uniform
        uint32_t wA = 0xffffffffu, wB = 0xffffffffu;
        if constexpr (DROP) {
          const int wi = kb * (KBF / 32) + 2 * pr + g;
          uint32_t mine;
          if constexpr (READY) {
            mine = pr == 0 ? kwc0 : kwc1;
          } else {
            mine = attn_keep_word(a.key, a.inner0, a.idx32, (uint64_t)(row_g * a.nw + wi), a.t16);
            if (a.keep && q_ok && wi < a.nw) a.keep[row_g * a.nw + wi] = mine;
          }
          halves(mine, wA, wB);                                           // the lower half's word (tile 2 pr) and the upper half's (2 pr + 1)
        }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          const int kt = 2 * pr + t2, key0 = K0 + 32 * kt;
          if (key0 >= kv_end) break;                                      // wave-uniform
          s_begin(kt);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) s_step(kt, ks);
          f32x16 s = sn;
          // ---- first V^T fragments on their way while the vector work runs
          const char* vimb = vim + 32 * kt * ROWB;
          auto o2_of = [&](int i) __attribute__((always_inline)) { return 16 * (i / DB) * ROWB + 64 * (i % DB); };
          bf16x8 vt = cat(lds_tr(vimb + trs_lo + o2_of(0)), lds_tr(vimb + trs_hi + o2_of(0)));
          // ---- scores -> probabilities (base 2); keys of this lane: key0 + 8 j + 4 g + r  <->  s[4 j + r]
          const bool edge = key0 + 32 > kv_lim;                           // wave-uniform
          if (edge) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
              const int key = key0 + 8 * (v >> 2) + 4 * g + (v & 3);
              float x = s[v] * cs;
              if (key >= Tk) x = -INFINITY;
              else if (key >= kv_lim) x += NEG_BIAS * unit;
              s[v] = x;
            }
          }
          float tmax = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
          for (int v = 3; v < 15; v += 2) tmax = fmaxf(fmaxf(tmax, s[v]), s[v + 1]);
          tmax = fmaxf(tmax, s[15]);
          if (!edge) tmax *= cs;
          {
            float t_lo, t_hi;
            halves(tmax, t_lo, t_hi);
            tmax = fmaxf(t_lo, t_hi);
          }
          const float m_new = fmaxf(m_run, tmax);
          const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // first tile: exp2(-inf) = 0
          m_run = m_new;
          const float sc = edge ? 1.f : cs;
          float lsum = 0.f;
          const uint32_t wg = (t2 == 0 ? wA : wB) >> (4 * g);
          if (!(ABL & 8) && __builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
              for (int v = 0; v < 16; ++v) oacc[db][v] *= alpha;
          }
          // probabilities in pairs: one v_cvt_pk_bf16_f32 per pair, and the dropout decisions applied to the PACKED pair - the
          // keep bits of a pair are neighbours in the word, so its mask is two bit-field extracts and one bit-field insert
          // (per element the compiler made three instructions of the mask and a cvt + perm of the packing: 71 vector
          // instructions per tile, now 40)
          typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
          u32x4 pk[2];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float x0 = __builtin_fmaf(s[2 * i], sc, -m_new), x1 = __builtin_fmaf(s[2 * i + 1], sc, -m_new);
            const float p0 = (ABL & 1) ? x0 : __builtin_amdgcn_exp2f(x0), p1 = (ABL & 1) ? x1 : __builtin_amdgcn_exp2f(x1);
            lsum += p0;
            lsum += p1;
            const bf16x2 pp = {(bf16_t)p0, (bf16_t)p1};
            uint32_t w2 = __builtin_bit_cast(uint32_t, pp);
            if constexpr (DROP) {
              const int pos = 8 * (i >> 1) + 2 * (i & 1);               // bit of element 2 i (element 2 i + 1: the next one)
              const uint32_t t1 = (uint32_t)__builtin_amdgcn_sbfe(wg, pos, 1), t2 = (uint32_t)__builtin_amdgcn_sbfe(wg, pos, 2);
              w2 &= __builtin_amdgcn_perm(t2, t1, 0x07060100u);          // low half from t1, high half from t2 (every bit of t2 above bit 0 is the second element's decision)
            }
            pk[i >> 2][i & 3] = w2;
          }
          l_run = __builtin_fmaf(l_run, alpha, lsum);
          bf16x8 pf[2];
          pf[0] = __builtin_bit_cast(bf16x8, pk[0]);
          pf[1] = __builtin_bit_cast(bf16x8, pk[1]);
          // ---- O^T += V^T.P^T
#pragma unroll
          for (int i = 0; i < 2 * DB; ++i) {
            bf16x8 nvt = vt;
            if (i + 1 < 2 * DB) nvt = cat(lds_tr(vimb + trs_lo + o2_of(i + 1)), lds_tr(vimb + trs_hi + o2_of(i + 1)));
            if constexpr ((ABL & 2) == 0) oacc[i % DB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt, pf[i / DB], oacc[i % DB], 0, 0, 0);
            else asm volatile("" ::"v"(vt), "v"(pf[i / DB]));
            vt = nvt;
          }
        }
      }
    }
    // ---- this slice is complete: O = O^T^T * keep_scale / l, lse = (m + log2 l) ln 2
    {
      float l_lo, l_hi;
      halves(l_run, l_lo, l_hi);
      const float l = l_lo + l_hi;
      const float inv = (DROP ? a.keep_scale : 1.f) / l;
      if (g == 0 && q_ok) a.lse[(int64_t)bh * Tq + q] = (m_run + __log2f(l)) * 0.6931471805599453f;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x16& x = oacc[db];
          const bf16x4 o4 = {(bf16_t)(x[4 * j] * inv), (bf16_t)(x[4 * j + 1] * inv), (bf16_t)(x[4 * j + 2] * inv), (bf16_t)(x[4 * j + 3] * inv)};
          *reinterpret_cast<bf16x4*>(ost + c * ROWB + (32 * db + 8 * j + 4 * g) * 2) = o4;
        }
      // the same wave reads back what it wrote (LDS operations of one wave complete in order): whole rows to memory
#pragma unroll
      for (int i = 0; i < QS * CH / 64; ++i) {
        const int n = lane + 64 * i, row = n / CH, ch = n % CH, qo = sl * QS + row;
        const bf16x8 val = *reinterpret_cast<const bf16x8*>(ost + row * ROWB + ch * 16);
        if (qo < Tq) *reinterpret_cast<bf16x8*>(a.o + (int64_t)b * a.bso + (int64_t)qo * a.ldo + h * DH + ch * 8) = val;
      }
    }
  }
}

}  // namespace a2

// ---- host side ---------------------------------------------------------------------------------------------------
// waves of the fused backward's workgroup: 4 (default) or 8.  Measured at T = 500, dh = 96, B H = 256 (tools/exp/attn_driver): eight
// waves 115 us without / 135 us with dropout against 118 / 118 us - every wave reads the whole Q / dO slice as its A operand, so
// twice the waves double those LDS reads, and the dropout variant no longer fits 256 registers without spills in the slice loop
static int g_opt_mha_bwd_waves = [] { const char* e = getenv("NEURST_MHA_BWD_WAVES"); return e && atoi(e) == 8 ? 8 : 4; }();
static int g_opt_fwd_ablate = 0;          // diagnostic: timing-only ablations of the wide forward (wrong results)
extern int g_opt_mha_small_fused;         // attention.hip
static int g_opt_mha_split = [] { const char* e = getenv("NEURST_MHA_SPLIT"); return e ? atoi(e) : 3; }();           // bit 1: forward, bit 2: backward; 0: one workgroup per (batch, head) whatever their number
static int g_opt_mha_fused = [] { const char* e = getenv("NEURST_MHA_FUSED"); return e ? atoi(e) : 1; }();           // 0: never, 1: encoder-size shapes, 2: whenever the kernel can take the shape
int neurst_attention2_set_option(const char* name, int value) {
  if (!strcmp(name, "mha_fused")) { g_opt_mha_fused = value; return 1; }
  if (!strcmp(name, "mha_split")) { g_opt_mha_split = value; return 1; }
  if (!strcmp(name, "mha_small_fused")) { g_opt_mha_small_fused = value; return 1; }
  if (!strcmp(name, "mha_fwd_ablate")) { g_opt_fwd_ablate = value; return 1; }
  if (!strcmp(name, "mha_bwd_waves")) { if (value != 4 && value != 8) return 0; g_opt_mha_bwd_waves = value; return 1; }
  return 0;
}


// workgroups per (batch, head) of the fused backward: the key blocks of a pair go to several workgroups when there are fewer
// pairs than CUs (NEURST_MHA_SPLIT=0 / neurst_set_option("mha_split", 0): always one)
static int64_t bwd_ksplit(int64_t B, int64_t H, int64_t Tk) {
  const int64_t nkb = cdiv64(Tk, a2::KB), per_chip = (g_opt_mha_split & 2) ? 256 / (B * H) : 1;
  return per_chip < 1 ? 1 : (per_chip > nkb ? nkb : per_chip);
}
static int64_t bwd_ws_part_floats(int64_t B, int64_t H, int64_t Tq, int64_t dh) { return B * H * cdiv64(Tq, a2::QS) * a2::QS * dh; }   // query rows padded to whole slices
extern "C" int64_t neurst_mha_bwd_workspace_bytes(int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t dh) {
  if (Tk <= a2::KB) return 0;
  // fp32 partial dQ per part (the largest split any option setting can ask for)
  const int64_t parts = std::max<int64_t>(1, std::min<int64_t>(cdiv64(Tk, a2::KB), 256 / (B * H) < 1 ? 1 : 256 / (B * H)));
  return parts * bwd_ws_part_floats(B, H, Tq, dh) * (int64_t)sizeof(float);
}

bool neurst_mha2_bwd_eligible(const Mha2Problem& p, int64_t ws_bytes) {
  if (!g_opt_mha_fused || p.dtype != NEURST_BF16) return false;
  if (p.dh != 96 && p.dh != 64) return false;
  if (p.mask_mode == NEURST_MASK_CAUSAL) return false;
  if (p.Tq > a2::MAX_TQ || p.B * p.H > 65535 * 16) return false;
  // one workgroup per (batch, head): below ~half a chip of them (single-utterance inference) the tiled kernels of
  // attention.hip, ceil(T / 64) workgroups per (batch, head), fill the CUs better
  // (short queries over long keys - the decoder's cross attention, Tq = 40 over Tk = 500 - stay here: 28 against 51 us per launch)
  // (with the key blocks of a pair on several workgroups the count that matters is workgroups: batch 8 = 64 pairs x 2 blocks)
  if (g_opt_mha_fused == 1 && (p.Tk < 128 || p.B * p.H * bwd_ksplit(p.B, p.H, p.Tk) < 128)) return false;
  if (ws_bytes < neurst_mha_bwd_workspace_bytes(p.B, p.H, p.Tq, p.Tk, p.dh)) return false;
  auto al16 = [](const void* x, int64_t ld) { return ((uintptr_t)x & 15) == 0 && (ld * 2) % 16 == 0; };
  if (!al16(p.q, p.ldq) || !al16(p.k, p.ldk) || !al16(p.v, p.ldv) || !al16(p.o, p.ldo) || !al16(p.d_o, p.ldo)) return false;
  if (!al16(p.dk, p.lddk) || !al16(p.dv, p.lddv) || !al16(p.dq, p.lddq)) return false;
  // 32-bit byte offsets inside one batch element's rows (LDS-DMA voffset)
  if (p.Tq * p.ldq * 2 >= (1ll << 31) || p.Tk * p.ldk * 2 >= (1ll << 31) || p.Tq * p.ldo * 2 >= (1ll << 31)) return false;
  if (p.rate > 0.f && !p.keep) return false;
  return true;
}

static long long* g_mha2_stamps = nullptr;
extern "C" void neurst_debug_mha2_stamps(long long* dev_buf) { g_mha2_stamps = dev_buf; }

int neurst_mha2_bwd_launch(const Mha2Problem& p, float* ws, hipStream_t s) {
  a2::Args a;
  memset(&a, 0, sizeof(a));
  a.stamps = g_mha2_stamps;
  a.q = (const bf16_t*)p.q; a.k = (const bf16_t*)p.k; a.v = (const bf16_t*)p.v; a.o = (const bf16_t*)p.o; a.d_o = (const bf16_t*)p.d_o;
  a.dq = (bf16_t*)p.dq; a.dk = (bf16_t*)p.dk; a.dv = (bf16_t*)p.dv;
  a.lse = p.lse; a.delta = p.delta; a.ws = ws; a.keep = p.rate > 0.f ? p.keep : nullptr; a.len_k = p.len_k;
  a.ldq = p.ldq; a.ldk = p.ldk; a.ldv = p.ldv; a.ldo = p.ldo; a.lddq = p.lddq; a.lddk = p.lddk; a.lddv = p.lddv;
  a.bsq = p.Tq * p.ldq; a.bsk = p.Tk * p.ldk; a.bsv = p.Tk * p.ldv; a.bso = p.Tq * p.ldo;
  a.B = (int)p.B; a.H = (int)p.H; a.Tq = (int)p.Tq; a.Tk = (int)p.Tk; a.nw = (int)cdiv64(p.Tk, 32);
  a.keypad = p.mask_mode == NEURST_MASK_KEYPAD;
  a.qscale = 1.0f / sqrtf((float)p.dh);
  a.keep_scale = keep_scale(p.rate);
  const int tqp = (int)cdiv64(p.Tq, a2::QS) * a2::QS;
  const size_t smem = p.dh == 96 ? a2::Lds<96>::bytes(tqp) : a2::Lds<64>::bytes(tqp);
  const bool w8 = g_opt_mha_bwd_waves == 8;
  auto kern = a.stamps ? (w8 ? a2::mha2_bwd_kernel<96, true, true, 8> : a2::mha2_bwd_kernel<96, true, true, 4>)
                       : (p.rate > 0.f ? (w8 ? a2::mha2_bwd_kernel<96, true, false, 8> : a2::mha2_bwd_kernel<96, true, false, 4>)
                                       : (w8 ? a2::mha2_bwd_kernel<96, false, false, 8> : a2::mha2_bwd_kernel<96, false, false, 4>));
  if (p.dh == 64) kern = p.rate > 0.f ? (w8 ? a2::mha2_bwd_kernel<64, true, false, 8> : a2::mha2_bwd_kernel<64, true, false, 4>)
                                      : (w8 ? a2::mha2_bwd_kernel<64, false, false, 8> : a2::mha2_bwd_kernel<64, false, false, 4>);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) { neurst_set_error("mha2_bwd: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e)); return NEURST_E_LAUNCH; }
  a.ksplit = (int)bwd_ksplit(p.B, p.H, p.Tk);
  a.ws_part = bwd_ws_part_floats(p.B, p.H, p.Tq, p.dh);
  kern<<<dim3((unsigned)(p.B * p.H * a.ksplit)), w8 ? 512 : 256, smem, s>>>(a);
  NEURST_CHECK_LAUNCH("mha2_bwd");
  if (a.ksplit > 1) {
    const int64_t units = p.B * p.H * p.Tq * (p.dh / 4);
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(units, 256 * 4), 2048);
    if (p.dh == 96) a2::mha2_dq_reduce_kernel<96><<<grid, 256, 0, s>>>(a);
    else a2::mha2_dq_reduce_kernel<64><<<grid, 256, 0, s>>>(a);
    NEURST_CHECK_LAUNCH("mha2_dq_reduce");
  }
  return NEURST_OK;
}

// ---- forward -------------------------------------------------------------------------------------------------------
// workgroups per (batch, head) of the wide forward: with fewer pairs than CUs (batch 16 at eight heads: 128) the groups of eight
// query slices of a pair go to several workgroups, each sweeping all keys for its own groups (K / V are read once more per
// part; the parts of a pair share an L2)
static int64_t fwd_qsplit(int64_t B, int64_t H, int64_t Tq) {
  const int64_t ngroup = cdiv64(cdiv64(Tq, a2::QS), a2::FW), per_chip = (g_opt_mha_split & 1) ? 256 / (B * H) : 1;
  return per_chip < 1 ? 1 : (per_chip > ngroup ? ngroup : per_chip);
}
bool neurst_mha2_fwd_eligible(const Mha2FwdProblem& p) {
  if (!g_opt_mha_fused || p.dtype != NEURST_BF16 || (p.dh != 96 && p.dh != 64)) return false;
  if (p.mask_mode == NEURST_MASK_CAUSAL) return false;
  if (p.B * p.H > 65535 * 16) return false;
  if (g_opt_mha_fused == 1 && (p.Tq < 128 || p.Tk < 128 || p.B * p.H * fwd_qsplit(p.B, p.H, p.Tq) < 128)) return false;   // as the backward
  auto al16 = [](const void* x, int64_t ld) { return ((uintptr_t)x & 15) == 0 && (ld * 2) % 16 == 0; };
  if (!al16(p.q, p.ldq) || !al16(p.k, p.ldk) || !al16(p.v, p.ldv) || !al16(p.o, p.ldo)) return false;
  if ((p.bsq * 2) % 16 || (p.bsk * 2) % 16 || (p.bsv * 2) % 16 || (p.bso * 2) % 16) return false;
  if (p.Tk * p.ldk * 2 >= (1ll << 31) || p.Tk * p.ldv * 2 >= (1ll << 31)) return false;   // 32-bit LDS-DMA offsets
  return true;
}
int neurst_mha2_fwd_launch(const Mha2FwdProblem& p, hipStream_t s) {
  a2::FwdArgs a;
  memset(&a, 0, sizeof(a));
  a.q = (const bf16_t*)p.q; a.k = (const bf16_t*)p.k; a.v = (const bf16_t*)p.v; a.o = (bf16_t*)p.o; a.lse = p.lse;
  a.keep = p.rate > 0.f ? p.keep : nullptr; a.keep_ready = a.keep ? p.keep_ready : 0; a.len_k = p.len_k;
  a.ldq = p.ldq; a.ldk = p.ldk; a.ldv = p.ldv; a.ldo = p.ldo; a.bsq = p.bsq; a.bsk = p.bsk; a.bsv = p.bsv; a.bso = p.bso;
  a.B = (int)p.B; a.H = (int)p.H; a.Tq = (int)p.Tq; a.Tk = (int)p.Tk; a.nw = (int)cdiv64(p.Tk, 32);
  a.keypad = p.mask_mode == NEURST_MASK_KEYPAD;
  a.key = p.key; a.inner0 = fmix32(p.key); a.t16 = p.rate > 0.f ? attn_threshold16(p.rate) : 0u;
  a.idx32 = (double)p.B * (double)p.H * (double)p.Tq * (double)a.nw * 4.0 < 4294967296.0;
  a.qscale = 1.0f / sqrtf((float)p.dh);
  a.keep_scale = keep_scale(p.rate);
  const size_t smem = p.dh == 96 ? a2::LdsF<96>::BYTES : a2::LdsF<64>::BYTES;
  const bool ready = a.keep_ready != 0;
  auto kern = p.rate > 0.f ? (ready ? a2::mha2_fwd_kernel<96, true, 0, true> : a2::mha2_fwd_kernel<96, true>) : a2::mha2_fwd_kernel<96, false>;
  if (p.dh == 64) kern = p.rate > 0.f ? (ready ? a2::mha2_fwd_kernel<64, true, 0, true> : a2::mha2_fwd_kernel<64, true>) : a2::mha2_fwd_kernel<64, false>;
  else switch (g_opt_fwd_ablate) {
    case 1: kern = a2::mha2_fwd_kernel<96, false, 1>; break;
    case 2: kern = a2::mha2_fwd_kernel<96, false, 2>; break;
    case 4: kern = a2::mha2_fwd_kernel<96, false, 4>; break;
    case 8: kern = a2::mha2_fwd_kernel<96, false, 8>; break;
    case 15: kern = a2::mha2_fwd_kernel<96, false, 15>; break;
    default: break;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) { neurst_set_error("mha2_fwd: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e)); return NEURST_E_LAUNCH; }
  a.qsplit = (int)fwd_qsplit(p.B, p.H, p.Tq);
  kern<<<dim3((unsigned)(p.B * p.H * a.qsplit)), 512, smem, s>>>(a);
  NEURST_CHECK_LAUNCH("mha2_fwd");
  return NEURST_OK;
}
