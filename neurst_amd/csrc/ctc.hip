// Loss-side kernels (HBM roofline): a11 CTC alpha-beta, a12 greedy CTC decode, a13 shrink,
// a17 label-smoothed cross entropy, a18 MSE, a2 loss mix.
//
// CTC follows tf.nn.ctc_loss (TF 1.15 ctc_loss_calculator; call site
// st/models/TransformerTPE.py:96-102): softmax inside, blank = V-1, labels >= blank end
// the label sequence, ctc_merge_repeated transitions, rows with U > T (or T == 0) give
// loss 0 / grad 0, rows without a valid path give loss +inf and grad = softmax.
// Traffic: logits are read ONCE from HBM (row cached in LDS), the gradient is written once
// (softmax * scale), then only the S = 2U+1 label columns per frame are touched again.
#include "common.h"

struct CtcWs {
  int32_t* lp;      // [B][Smax]  l' (labels with blanks)
  int32_t* next;    // [B][Smax]  next state with the same symbol, -1 = none
  int32_t* first;   // [B][Smax]  1 if no earlier state has the same symbol
  int32_t* meta;    // [B][4]     S, active(1: run recursion), nopath, unused
  double* logy;     // [B][T][Smax]  fp64 emission probabilities y_t(l'_u) = softmax(logits_t)[l'_u] (name kept from the log-space form)
  double* alpha;    // [B][T][Smax]  overwritten by the posteriors
};
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static CtcWs carve_ws(void* ws, int64_t B, int64_t T, int64_t Smax) {
  char* p = reinterpret_cast<char*>(ws);
  CtcWs w;
  w.lp = reinterpret_cast<int32_t*>(p); p += align256(sizeof(int32_t) * B * Smax);
  w.next = reinterpret_cast<int32_t*>(p); p += align256(sizeof(int32_t) * B * Smax);
  w.first = reinterpret_cast<int32_t*>(p); p += align256(sizeof(int32_t) * B * Smax);
  w.meta = reinterpret_cast<int32_t*>(p); p += align256(sizeof(int32_t) * B * 4);
  w.logy = reinterpret_cast<double*>(p); p += align256(sizeof(double) * B * T * Smax);
  w.alpha = reinterpret_cast<double*>(p);
  return w;
}
extern "C" int64_t neurst_ctc_workspace_bytes(int64_t B, int64_t T, int64_t Umax) {
  int64_t Smax = 2 * Umax + 1;
  return (int64_t)(3 * align256(sizeof(int32_t) * B * Smax) + align256(sizeof(int32_t) * B * 4) +
                   2 * align256(sizeof(double) * B * T * Smax));
}

// ---- prep: one thread block per utterance builds l', the same-symbol chains and the feasibility flags
__global__ void ctc_prep_kernel(const int32_t* __restrict__ labels, const int32_t* __restrict__ len_labels,
                                const int32_t* __restrict__ len_logits, CtcWs w, int64_t Umax, int64_t Smax, int64_t T, int V) {
  const int64_t b = blockIdx.x;
  int32_t* lp = w.lp + b * Smax;
  int32_t* nx = w.next + b * Smax;
  int32_t* fi = w.first + b * Smax;
  __shared__ int Ueff;
  const int blank = V - 1;
  int U = len_labels[b];
  U = U < 0 ? 0 : (U > Umax ? (int)Umax : U);
  int Tb = len_logits[b];
  Tb = Tb < 0 ? 0 : (Tb > T ? (int)T : Tb);
  if (threadIdx.x == 0) {
    int n = 0;
    while (n < U && labels[b * Umax + n] < blank && labels[b * Umax + n] >= 0) ++n;
    Ueff = n;
  }
  __syncthreads();
  const int S = 2 * Ueff + 1;
  for (int u = threadIdx.x; u < Smax; u += blockDim.x) lp[u] = (u < S && (u & 1)) ? labels[b * Umax + (u >> 1)] : blank;
  __syncthreads();
  for (int u = threadIdx.x; u < Smax; u += blockDim.x) {
    int nxt = -1, first = 1;
    if (u < S) {
      for (int k = u + 1; k < S; ++k) if (lp[k] == lp[u]) { nxt = k; break; }
      for (int k = 0; k < u; ++k) if (lp[k] == lp[u]) { first = 0; break; }
    }
    nx[u] = nxt;
    fi[u] = first;
  }
  if (threadIdx.x == 0) {
    w.meta[b * 4 + 0] = S;
    w.meta[b * 4 + 1] = (Tb > 0 && U <= Tb) ? 1 : 0;   // ignore_longer_outputs_than_inputs / empty sequence
    w.meta[b * 4 + 2] = 0;
  }
}

// first-maximum argmax + online log-sum-exp over one row cached in LDS
struct RowStat { float m, s; int arg; };
__device__ __forceinline__ RowStat combine(RowStat a, RowStat b) {
  RowStat r;
  r.m = fmaxf(a.m, b.m);
  r.s = (a.m == -INFINITY ? 0.f : a.s * expf(a.m - r.m)) + (b.m == -INFINITY ? 0.f : b.s * expf(b.m - r.m));
  // argmax of the larger maximum, smallest index on ties
  r.arg = (a.m > b.m) ? a.arg : ((b.m > a.m) ? b.arg : (a.arg < b.arg ? a.arg : b.arg));
  return r;
}
__device__ __forceinline__ RowStat block_rowstat(const float* __restrict__ row, float* __restrict__ cache, int V, RowStat* red) {
  RowStat st = {-INFINITY, 0.f, 0x7fffffff};
  for (int i = threadIdx.x; i < V; i += blockDim.x) {
    float x = row[i];
    if (cache) cache[i] = x;
    if (x > st.m) { st.s = (st.m == -INFINITY ? 0.f : st.s * expf(st.m - x)) + 1.f; st.m = x; st.arg = i; }
    else st.s += expf(x - st.m);
  }
  for (int o = 32; o > 0; o >>= 1) {
    RowStat other = {__shfl_xor(st.m, o, 64), __shfl_xor(st.s, o, 64), __shfl_xor(st.arg, o, 64)};
    st = combine(st, other);
  }
  const int wv = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wv] = st;
  __syncthreads();
  RowStat r = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = combine(r, red[i]);
  return r;
}

// ---- A: per frame softmax (written as scaled gradient base), argmax and gathered log y
template <typename TG>
__global__ void __launch_bounds__(256) ctc_frame_kernel(const float* __restrict__ logits, const int32_t* __restrict__ len_logits,
                                                       CtcWs w, TG* __restrict__ grad, int32_t* __restrict__ argmax, int64_t T, int V,
                                                       int64_t Smax, float grad_scale, int use_cache) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* cache = use_cache ? reinterpret_cast<float*>(smem) : nullptr;
  __shared__ RowStat red[4];
  const int64_t bt = blockIdx.x, b = bt / T, t = bt % T;
  const float* row = logits + bt * V;
  TG* grow = grad + bt * V;
  int Tb = len_logits[b];
  Tb = Tb > T ? (int)T : Tb;
  if (t >= Tb) {
    for (int i = threadIdx.x; i < V; i += blockDim.x) grow[i] = from_f32<TG>(0.f);
    if (threadIdx.x == 0 && argmax) argmax[bt] = V - 1;
    return;
  }
  RowStat st = block_rowstat(row, cache, V, red);
  if (threadIdx.x == 0 && argmax) argmax[bt] = st.arg;
  const bool active = w.meta[b * 4 + 1] != 0;
  const float inv = 1.f / st.s;
  const float* src = cache ? cache : row;
  if (active) {
    const int S = w.meta[b * 4 + 0];
    const int32_t* lp = w.lp + b * Smax;
    double* ly = w.logy + bt * Smax;
    const double lse = (double)st.m + log((double)st.s);
    for (int u = threadIdx.x; u < S; u += blockDim.x) ly[u] = exp((double)src[lp[u]] - lse);
  }
  // in-place safe: every element is read (from the LDS cache or by the same thread) before it is written
  for (int i = threadIdx.x; i < V; i += blockDim.x) {
    float y = active ? expf(src[i] - st.m) * inv * grad_scale : 0.f;
    grow[i] = from_f32<TG>(y);
  }
}


// ---- A (register-resident rows): V <= 256*4*NV4, V % 4 == 0, 16-byte aligned rows.  Each thread keeps its
// share of the row in registers: one HBM read, one exp per element, one vector store of the gradient base.
template <typename TG, int NV4>
__global__ void __launch_bounds__(256) ctc_frame_reg_kernel(const float* __restrict__ logits, const int32_t* __restrict__ len_logits,
                                                           CtcWs w, TG* __restrict__ grad, int32_t* __restrict__ argmax, int64_t T, int V,
                                                           int64_t Smax, float grad_scale) {
  __shared__ RowStat red[4];
  __shared__ float reds[4];
  const int64_t bt = blockIdx.x, b = bt / T, t = bt % T;
  const int nv4 = V >> 2, tid = threadIdx.x;
  const f32x4* row4 = reinterpret_cast<const f32x4*>(logits + bt * V);
  TG* grow = grad + bt * V;
  auto store4 = [&](int idx, f32x4 y) {
    if constexpr (sizeof(TG) == 4) *reinterpret_cast<f32x4*>(grow + 4 * idx) = y;
    else *reinterpret_cast<bf16x4*>(grow + 4 * idx) = bf16x4{(bf16_t)y[0], (bf16_t)y[1], (bf16_t)y[2], (bf16_t)y[3]};
  };
  int Tb = len_logits[b];
  Tb = Tb > T ? (int)T : Tb;
  if (t >= Tb) {
    for (int idx = tid; idx < nv4; idx += 256) store4(idx, f32x4{0.f, 0.f, 0.f, 0.f});
    if (tid == 0 && argmax) argmax[bt] = V - 1;
    return;
  }
  f32x4 x[NV4];
  RowStat st = {-INFINITY, 0.f, 0x7fffffff};
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int idx = tid + i * 256;
    if (idx < nv4) {
      x[i] = row4[idx];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (x[i][e] > st.m) { st.m = x[i][e]; st.arg = 4 * idx + e; }
    } else {
      x[i] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(st.m, o, 64);
    const int oa = __shfl_xor(st.arg, o, 64);
    if (om > st.m || (om == st.m && oa < st.arg)) { st.m = om; st.arg = oa; }
  }
  if ((tid & 63) == 0) red[tid >> 6] = st;
  __syncthreads();
  st = red[0];
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (red[i].m > st.m || (red[i].m == st.m && red[i].arg < st.arg)) st = red[i];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { x[i][e] = expf(x[i][e] - st.m); sum += x[i][e]; }   // exp(-inf) = 0 for the unused tail
  sum = wave_sum(sum);
  if ((tid & 63) == 0) reds[tid >> 6] = sum;
  __syncthreads();
  sum = reds[0] + reds[1] + reds[2] + reds[3];
  if (tid == 0 && argmax) argmax[bt] = st.arg;
  const bool active = w.meta[b * 4 + 1] != 0;
  if (active) {
    const int S = w.meta[b * 4 + 0];
    const int32_t* lp = w.lp + b * Smax;
    double* ly = w.logy + bt * Smax;
    const double lse = (double)st.m + log((double)sum);
    const float* row = logits + bt * V;
    for (int u = tid; u < S; u += 256) ly[u] = exp((double)row[lp[u]] - lse);
  }
  __syncthreads();   // in-place gradient: the label columns above are read before any element is overwritten
  const float k = active ? grad_scale / sum : 0.f;
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int idx = tid + i * 256;
    if (idx < nv4) store4(idx, x[i] * k);
  }
}

// ---- B: alpha / beta recursion, ONE wavefront per utterance.
// Lane l owns the NPL consecutive states l*NPL .. l*NPL+NPL-1 of l' (S = 2U+1 <= 64 NPL), so a frame's update needs two values of
// the previous (alpha) or next (beta) lane, moved with DPP wave shifts: no LDS, no barrier.  A log-space recursion (round 1:
// one thread per state, fp64, 986 us at T = 500) spends its time in three exp and one log per state and frame; here the
// arithmetic is LINEAR fp64 with rescaling - alpha_t(u) =
// y_t(u) (alpha(u) + alpha(u-1) + [skip] alpha(u-2)) on probabilities y (written by the frame kernel), every 4th frame divided
// by the power of two below the wave's sum (exact; the exponents add up to log2 of the scale of p).  All terms are positive, so there is no cancellation: the
// result agrees with the log-space recursion to ~1e-13.  The posteriors need no scale at all: alpha_t(u) beta_t(u) summed
// over u is p(l|x) times the frame's scale for EVERY t (beta excludes the emission at t), so post_t(u) = alpha beta / sum_u alpha beta.
__device__ __forceinline__ double dpp_f64(double x, double fill, const int ctrl_is_shl) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  const int flo = __double2loint(fill), fhi = __double2hiint(fill);
  if (ctrl_is_shl) {   // value of lane l + 1 (lane 63 keeps `fill`)
    lo = __builtin_amdgcn_update_dpp(flo, lo, 0x130, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(fhi, hi, 0x130, 0xF, 0xF, false);
  } else {             // value of lane l - 1 (lane 0 keeps `fill`)
    lo = __builtin_amdgcn_update_dpp(flo, lo, 0x138, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(fhi, hi, 0x138, 0xF, 0xF, false);
  }
  return __hiloint2double(hi, lo);
}
// wave-wide fp64 sum over the DPP network (the six steps of wave_sum_dpp in common.h on both halves of the double): a few
// cycles per step, against ~100 of latency for each of the twelve LDS-crossbar shuffles of a __shfl_xor butterfly
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_step_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWS, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWS, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_step_f64<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v += dpp_step_f64<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v += dpp_step_f64<0x141, 0xF>(v);   // row_half_mirror
  v += dpp_step_f64<0x140, 0xF>(v);   // row_mirror
  v += dpp_step_f64<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_step_f64<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

template <int NPL>
__global__ void __launch_bounds__(64) ctc_alpha_beta_wave_kernel(const int32_t* __restrict__ len_logits, CtcWs w, float* __restrict__ loss, int64_t T,
                                                                  int64_t Smax) {
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  const int S = w.meta[b * 4 + 0];
  int Tb = len_logits[b];
  Tb = Tb > T ? (int)T : Tb;
  if (!w.meta[b * 4 + 1]) { if (lane == 0) loss[b] = 0.f; return; }
  const int32_t* lp = w.lp + b * Smax;
  const double* yv = w.logy + b * T * Smax;         // probabilities y_t(u) (linear), written by the frame kernel
  double* alpha = w.alpha + b * T * Smax;
  const int u0 = lane * NPL;
  const int blank = lp[0];
  bool skip_in[NPL], skip_out[NPL], live[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int u = u0 + i;
    live[i] = u < S;
    skip_in[i] = live[i] && u >= 2 && lp[u] != blank && lp[u] != lp[u - 2];
    skip_out[i] = live[i] && u + 2 < S && lp[u + 2] != blank && lp[u + 2] != lp[u];
  }
  auto load_row = [&](const double* row, double (&v)[NPL]) {
#pragma unroll
    for (int i = 0; i < NPL; ++i) v[i] = live[i] ? row[u0 + i] : 0.0;
  };
  // Frames are fetched BLK at a time, one block ahead of the recursion: a frame's update is ~100 cycles of dependent
  // fp64 arithmetic, a global load ~1000 - fetched frame by frame (first version: 717 us at T = 500) the loop only waits.
  constexpr int BLK = NPL <= 2 ? 8 : (NPL <= 4 ? 4 : (NPL <= 8 ? 2 : 1));
  auto load_block = [&](const double* base, int t0, int step, double (&v)[BLK][NPL]) {   // frames t0, t0 + step, ... (inside [0, Tb))
#pragma unroll
    for (int j = 0; j < BLK; ++j) {
      const int t = t0 + j * step;
      const bool ok = t >= 0 && t < Tb;
#pragma unroll
      for (int i = 0; i < NPL; ++i) v[j][i] = (ok && live[i]) ? base[(int64_t)t * Smax + u0 + i] : 0.0;
    }
  };
  // ---------------- alpha
  double a[NPL], y[BLK][NPL], yn[BLK][NPL];
  {
    double y0[NPL];
    load_row(yv, y0);
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      a[i] = (live[i] && u0 + i < 2) ? y0[i] : 0.0;
      if (live[i]) alpha[u0 + i] = a[i];
    }
  }
  int64_t scale_exp = 0;                 // log2 of the product of all divisors
  load_block(yv, 1, 1, y);
  for (int tb = 1; tb < Tb; tb += BLK) {
    load_block(yv, tb + BLK, 1, yn);
#pragma unroll
    for (int j = 0; j < BLK; ++j) {
      const int t = tb + j;
      if (t < Tb) {
        const double pm1 = dpp_f64(a[NPL - 1], 0.0, 0);                         // state u0 - 1
        const double pm2 = dpp_f64(NPL >= 2 ? a[NPL - 2] : 0.0, 0.0, 0);        // state u0 - 2
        double na[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
          const double m1 = i >= 1 ? a[i - 1] : pm1;
          const double m2 = i >= 2 ? a[i - 2] : (i == 1 ? pm1 : pm2);
          na[i] = y[j][i] * (a[i] + m1 + (skip_in[i] ? m2 : 0.0));
        }
        if ((t & 3) == 0) {                                                      // rescale: keeps the row near 1
          double sum = 0.0;
#pragma unroll
          for (int i = 0; i < NPL; ++i) sum += na[i];
          sum = wave_sum_f64(sum);
          if (sum > 0.0) {                                                       // by the power of two below the sum: exact, no
            const int e = __builtin_amdgcn_frexp_exp(sum);                       // division or logarithm on the recurrence
#pragma unroll
            for (int i = 0; i < NPL; ++i) na[i] = ldexp(na[i], -e);
            scale_exp += e;
          }
        }
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
          a[i] = na[i];
          if (live[i]) alpha[(int64_t)t * Smax + u0 + i] = a[i];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < BLK; ++j)
#pragma unroll
      for (int i = 0; i < NPL; ++i) y[j][i] = yn[j][i];
  }
  // p(l | x) = alpha_{T-1}(S-1) + alpha_{T-1}(S-2)
  double tail = 0.0;
#pragma unroll
  for (int i = 0; i < NPL; ++i)
    if (live[i] && u0 + i >= S - 2) tail += a[i];
  tail = wave_sum_f64(tail);
  const bool nopath = !(tail > 0.0);
  if (lane == 0) {
    loss[b] = nopath ? INFINITY : (float)(-((double)scale_exp * 0.6931471805599453094 + log(tail)));
    if (nopath) w.meta[b * 4 + 2] = 1;     // no valid path: gradient stays softmax
  }
  if (nopath) return;
  // ---------------- beta (emission at t excluded) and posteriors, written over alpha
  double be[NPL], at[BLK][NPL], atn[BLK][NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) be[i] = (live[i] && u0 + i >= S - 2) ? 1.0 : 0.0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's own alpha stores, re-read below
  load_block(yv, Tb - 1, -1, y);
  load_block(alpha, Tb - 1, -1, at);
  for (int tb = Tb - 1; tb >= 0; tb -= BLK) {
    load_block(yv, tb - BLK, -1, yn);
    load_block(alpha, tb - BLK, -1, atn);
#pragma unroll
    for (int j = 0; j < BLK; ++j) {
      const int t = tb - j;
      if (t >= 0) {
        // posterior of frame t
        double ab[NPL], sum = 0.0;
#pragma unroll
        for (int i = 0; i < NPL; ++i) { ab[i] = at[j][i] * be[i]; sum += ab[i]; }
        sum = wave_sum_f64(sum);
        const double inv = sum > 0.0 ? 1.0 / sum : 0.0;
#pragma unroll
        for (int i = 0; i < NPL; ++i)
          if (live[i]) alpha[(int64_t)t * Smax + u0 + i] = ab[i] * inv;
        // beta_{t-1}(u) = z(u) + z(u+1) + [skip] z(u+2), z = beta_t y_t
        double z[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i) z[i] = be[i] * y[j][i];
        const double np1 = dpp_f64(z[0], 0.0, 1);                                // state u0 + NPL
        const double np2 = dpp_f64(NPL >= 2 ? z[1] : 0.0, 0.0, 1);               // state u0 + NPL + 1
        double nb[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
          const double p1 = i + 1 < NPL ? z[i + 1] : np1;
          const double p2 = i + 2 < NPL ? z[i + 2] : (i + 2 == NPL ? np1 : np2);
          nb[i] = z[i] + p1 + (skip_out[i] ? p2 : 0.0);
        }
        if ((t & 3) == 0) {
          double s2 = 0.0;
#pragma unroll
          for (int i = 0; i < NPL; ++i) s2 += nb[i];
          s2 = wave_sum_f64(s2);
          if (s2 > 0.0) {
            const int e = __builtin_amdgcn_frexp_exp(s2);
#pragma unroll
            for (int i = 0; i < NPL; ++i) nb[i] = ldexp(nb[i], -e);
          }
        }
#pragma unroll
        for (int i = 0; i < NPL; ++i) be[i] = nb[i];
      }
    }
#pragma unroll
    for (int j = 0; j < BLK; ++j)
#pragma unroll
      for (int i = 0; i < NPL; ++i) { y[j][i] = yn[j][i]; at[j][i] = atn[j][i]; }
  }
}

// ---- C: grad[b,t,l'_u] -= scale * sum of the posteriors of all states emitting that symbol
template <typename TG>
__global__ void ctc_grad_fix_kernel(const int32_t* __restrict__ len_logits, CtcWs w, TG* __restrict__ grad, int64_t T, int V,
                                    int64_t Smax, float grad_scale) {
  const int64_t bt = blockIdx.x, b = bt / T, t = bt % T;
  if (!w.meta[b * 4 + 1] || w.meta[b * 4 + 2]) return;
  int Tb = len_logits[b];
  if (t >= Tb) return;
  const int S = w.meta[b * 4 + 0];
  const int32_t* lp = w.lp + b * Smax;
  const int32_t* nx = w.next + b * Smax;
  const int32_t* fi = w.first + b * Smax;
  const double* post = w.alpha + bt * Smax;
  TG* grow = grad + bt * V;
  for (int u = threadIdx.x; u < S; u += blockDim.x) {
    if (!fi[u]) continue;
    double s = 0.0;
    for (int k = u; k >= 0; k = nx[k]) s += post[k];
    grow[lp[u]] = from_f32<TG>(to_f32<TG>(grow[lp[u]]) - (float)s * grad_scale);
  }
}

// phases: 1 = per-frame part (softmax as gradient base, arg-max, gathered log-probabilities), 2 = alpha-beta recursion and
// the sparse gradient correction; 3 = both.  Phase 2 may run on another stream once phase 1 has finished there.
extern "C" int neurst_ctc_loss_fwd_bwd_phased(const float* logits, const int32_t* len_logits, const int32_t* labels,
                                              const int32_t* len_labels, float* loss, void* grad, int grad_dtype, int32_t* argmax,
                                              void* workspace, int64_t B, int64_t T, int64_t V, int64_t Umax, float grad_scale,
                                              int phases, void* stream) {
  NEURST_CHECK_ARG(phases >= 1 && phases <= 3, "ctc: phases must be 1, 2 or 3");
  NEURST_CHECK_ARG(logits && len_logits && labels && len_labels && loss && grad && workspace, "ctc: null pointer");
  NEURST_CHECK_ARG(B > 0 && T > 0 && V > 1 && Umax > 0, "ctc: bad shape");
  const int64_t Smax = 2 * Umax + 1;
  NEURST_CHECK_ARG(Smax <= 1024, "ctc: label length %ld too long (2U+1 must be <= 1024)", (long)Umax);
  NEURST_CHECK_ARG(grad_dtype == NEURST_F32 || (const void*)grad != (const void*)logits, "ctc: bf16 gradient cannot alias the logits");
  hipStream_t s = as_stream(stream);
  CtcWs w = carve_ws(workspace, B, T, Smax);
  if (phases & 1) {
  ctc_prep_kernel<<<(int)B, 128, 0, s>>>(labels, len_labels, len_logits, w, Umax, Smax, T, (int)V);
  NEURST_CHECK_LAUNCH("ctc_prep");
  const int use_cache = V * sizeof(float) <= 64 * 1024;
  const size_t shm = use_cache ? V * sizeof(float) : 0;
  NEURST_CHECK_ARG(use_cache || (const void*)grad != (const void*)logits, "ctc: in-place gradient needs V*4 <= 64 KiB");
  const bool vec = V % 4 == 0 && ((uintptr_t)logits & 15) == 0 && ((uintptr_t)grad & 15) == 0 && V <= 256 * 4 * 16;
  const unsigned nblk = (unsigned)(B * T);
#define CTC_FRAME_REG(TG, NV4) ctc_frame_reg_kernel<TG, NV4><<<nblk, 256, 0, s>>>(logits, len_logits, w, (TG*)grad, argmax, T, (int)V, Smax, grad_scale)
  if (vec && grad_dtype == NEURST_BF16) {
    if (V <= 256 * 4 * 3) CTC_FRAME_REG(bf16_t, 3); else if (V <= 256 * 4 * 10) CTC_FRAME_REG(bf16_t, 10); else CTC_FRAME_REG(bf16_t, 16);
  } else if (vec) {
    if (V <= 256 * 4 * 3) CTC_FRAME_REG(float, 3); else if (V <= 256 * 4 * 10) CTC_FRAME_REG(float, 10); else CTC_FRAME_REG(float, 16);
  } else if (grad_dtype == NEURST_BF16)
    ctc_frame_kernel<bf16_t><<<nblk, 256, shm, s>>>(logits, len_logits, w, (bf16_t*)grad, argmax, T, (int)V, Smax, grad_scale, use_cache);
  else
    ctc_frame_kernel<float><<<nblk, 256, shm, s>>>(logits, len_logits, w, (float*)grad, argmax, T, (int)V, Smax, grad_scale, use_cache);
#undef CTC_FRAME_REG
  NEURST_CHECK_LAUNCH("ctc_frame");
  }
  if (!(phases & 2)) return NEURST_OK;
  const int npl = (int)((Smax + 63) / 64);
  if (npl <= 2) ctc_alpha_beta_wave_kernel<2><<<(int)B, 64, 0, s>>>(len_logits, w, loss, T, Smax);
  else if (npl <= 4) ctc_alpha_beta_wave_kernel<4><<<(int)B, 64, 0, s>>>(len_logits, w, loss, T, Smax);
  else if (npl <= 8) ctc_alpha_beta_wave_kernel<8><<<(int)B, 64, 0, s>>>(len_logits, w, loss, T, Smax);
  else ctc_alpha_beta_wave_kernel<16><<<(int)B, 64, 0, s>>>(len_logits, w, loss, T, Smax);
  NEURST_CHECK_LAUNCH("ctc_alpha_beta");
  if (grad_dtype == NEURST_BF16)
    ctc_grad_fix_kernel<bf16_t><<<(unsigned)(B * T), 128, 0, s>>>(len_logits, w, (bf16_t*)grad, T, (int)V, Smax, grad_scale);
  else
    ctc_grad_fix_kernel<float><<<(unsigned)(B * T), 128, 0, s>>>(len_logits, w, (float*)grad, T, (int)V, Smax, grad_scale);
  NEURST_CHECK_LAUNCH("ctc_grad_fix");
  return NEURST_OK;
}
extern "C" int neurst_ctc_loss_fwd_bwd(const float* logits, const int32_t* len_logits, const int32_t* labels, const int32_t* len_labels,
                                       float* loss, void* grad, int grad_dtype, int32_t* argmax, void* workspace, int64_t B, int64_t T,
                                       int64_t V, int64_t Umax, float grad_scale, void* stream) {
  return neurst_ctc_loss_fwd_bwd_phased(logits, len_logits, labels, len_labels, loss, grad, grad_dtype, argmax, workspace, B, T, V, Umax,
                                        grad_scale, 3, stream);
}

// ---- per-row first-maximum argmax (inference)
__global__ void __launch_bounds__(256) argmax_rows_kernel(const float* __restrict__ logits, int32_t* __restrict__ argmax, int V) {
  __shared__ RowStat red[4];
  RowStat st = block_rowstat(logits + (int64_t)blockIdx.x * V, nullptr, V, red);
  if (threadIdx.x == 0) argmax[blockIdx.x] = st.arg;
}
extern "C" int neurst_argmax_rows(const float* logits, int32_t* argmax, int64_t rows, int64_t V, void* stream) {
  NEURST_CHECK_ARG(rows > 0 && V > 0, "argmax_rows: bad shape");
  argmax_rows_kernel<<<(unsigned)rows, 256, 0, as_stream(stream)>>>(logits, argmax, (int)V);
  NEURST_CHECK_LAUNCH("argmax_rows");
  return NEURST_OK;
}

// ---------------------------------------------------------------- a12 / a13: keep flags + compaction
// index[b][j] = source frame of the j-th kept frame, -1 beyond the count
__global__ void __launch_bounds__(256) keep_index_kernel(const int32_t* __restrict__ argmax, const int32_t* __restrict__ lens,
                                                        int32_t* __restrict__ index, int32_t* __restrict__ new_len, int64_t T,
                                                        int32_t blank) {
  __shared__ int counts[256];
  const int64_t b = blockIdx.x;
  const int32_t* am = argmax + b * T;
  int32_t* idx = index + b * T;
  int len = lens[b];
  len = len > T ? (int)T : len;
  const int per = (int)((T + 255) / 256);
  const int t0 = threadIdx.x * per, t1 = (t0 + per < T) ? t0 + per : (int)T;
  int c = 0;
  for (int t = t0; t < t1; ++t) {
    int prev = t == 0 ? blank : am[t - 1];
    c += (t < len && am[t] != blank && am[t] != prev) ? 1 : 0;
  }
  counts[threadIdx.x] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) { int x = counts[i]; counts[i] = run; run += x; }
    new_len[b] = run;
  }
  __syncthreads();
  int pos = counts[threadIdx.x];
  for (int t = t0; t < t1; ++t) {
    int prev = t == 0 ? blank : am[t - 1];
    if (t < len && am[t] != blank && am[t] != prev) idx[pos++] = t;
  }
  __syncthreads();
  const int total = new_len[b];
  for (int j = total + threadIdx.x; j < T; j += blockDim.x) idx[j] = -1;
}
__global__ void greedy_gather_kernel(const int32_t* __restrict__ argmax, int32_t* __restrict__ out, int64_t B, int64_t T) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  int64_t b = i / T;
  int src = out[i];
  out[i] = src >= 0 ? argmax[b * T + src] : 0;
}
extern "C" int neurst_ctc_greedy(const int32_t* argmax, const int32_t* len_logits, int32_t* out, int32_t* out_len, int64_t B, int64_t T,
                                 int32_t blank, void* stream) {
  NEURST_CHECK_ARG(B > 0 && T > 0, "ctc_greedy: bad shape");
  keep_index_kernel<<<(int)B, 256, 0, as_stream(stream)>>>(argmax, len_logits, out, out_len, T, blank);
  NEURST_CHECK_LAUNCH("ctc_greedy_index");
  greedy_gather_kernel<<<(int)cdiv64(B * T, 256), 256, 0, as_stream(stream)>>>(argmax, out, B, T);
  NEURST_CHECK_LAUNCH("ctc_greedy_gather");
  return NEURST_OK;
}
extern "C" int neurst_shrink_index(const int32_t* argmax, const int32_t* lens, int32_t* index, int32_t* new_len, int64_t B, int64_t T,
                                   int32_t blank, void* stream) {
  NEURST_CHECK_ARG(B > 0 && T > 0, "shrink_index: bad shape");
  keep_index_kernel<<<(int)B, 256, 0, as_stream(stream)>>>(argmax, lens, index, new_len, T, blank);
  NEURST_CHECK_LAUNCH("shrink_index");
  return NEURST_OK;
}

template <typename T>
__global__ void shrink_gather_kernel(const T* __restrict__ x, const int32_t* __restrict__ index, T* __restrict__ out, int64_t B, int64_t Tn,
                                     int64_t Tout, int64_t D) {
  const int64_t total = B * Tout * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t d = i % D, r = i / D;
    int64_t j = r % Tout, b = r / Tout;
    int src = index[b * Tn + j];
    out[i] = src >= 0 ? x[(b * Tn + src) * D + d] : from_f32<T>(0.f);
  }
}
template <typename T>
__global__ void shrink_scatter_kernel(const T* __restrict__ dout, const int32_t* __restrict__ index, T* __restrict__ dx, int64_t B, int64_t Tn,
                                      int64_t Tout, int64_t D) {
  const int64_t total = B * Tout * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t d = i % D, r = i / D;
    int64_t j = r % Tout, b = r / Tout;
    int src = index[b * Tn + j];
    if (src >= 0) dx[(b * Tn + src) * D + d] = dout[i];
  }
}
static inline int sgrid(int64_t n) { int64_t g = cdiv64(n, 256); return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }
extern "C" int neurst_shrink_fwd(const void* x, const int32_t* index, void* out, int64_t B, int64_t T, int64_t Tout, int64_t D, int dtype,
                                 void* stream) {
  NEURST_CHECK_ARG(B > 0 && T > 0 && Tout > 0 && Tout <= T && D > 0, "shrink_fwd: bad shape");
  if (dtype == NEURST_BF16) shrink_gather_kernel<bf16_t><<<sgrid(B * Tout * D), 256, 0, as_stream(stream)>>>((const bf16_t*)x, index, (bf16_t*)out, B, T, Tout, D);
  else shrink_gather_kernel<float><<<sgrid(B * Tout * D), 256, 0, as_stream(stream)>>>((const float*)x, index, (float*)out, B, T, Tout, D);
  NEURST_CHECK_LAUNCH("shrink_fwd");
  return NEURST_OK;
}
extern "C" int neurst_shrink_bwd(const void* dout, const int32_t* index, void* dx, int64_t B, int64_t T, int64_t Tout, int64_t D, int dtype,
                                 void* stream) {
  NEURST_CHECK_ARG(B > 0 && T > 0 && Tout > 0 && Tout <= T && D > 0, "shrink_bwd: bad shape");
  hipError_t e = hipMemsetAsync(dx, 0, (size_t)(B * T * D) * dtype_size(dtype), as_stream(stream));
  if (e != hipSuccess) { neurst_set_error("shrink_bwd: memset failed: %s", hipGetErrorString(e)); return NEURST_E_LAUNCH; }
  if (dtype == NEURST_BF16) shrink_scatter_kernel<bf16_t><<<sgrid(B * Tout * D), 256, 0, as_stream(stream)>>>((const bf16_t*)dout, index, (bf16_t*)dx, B, T, Tout, D);
  else shrink_scatter_kernel<float><<<sgrid(B * Tout * D), 256, 0, as_stream(stream)>>>((const float*)dout, index, (float*)dx, B, T, Tout, D);
  NEURST_CHECK_LAUNCH("shrink_bwd");
  return NEURST_OK;
}

// Speculative sizing of the shrunk tensor (no reference counterpart: st/models/tools/blocks.py:8-38 sizes it by a
// device -> host read of the counts, which stalls the host until the acoustic encoder has run): the host allocates for a
// bound it guesses from earlier steps; this kernel clamps the counts to the bound (every later kernel reads lengths from
// the device), reports the true maximum and raises a flag when a count did not fit - the guarded Adam entry point then
// leaves the parameters alone and the host repeats the batch with the full length.  state[0] = overflow, state[1] = max.
__global__ void __launch_bounds__(256) shrink_clamp_kernel(int32_t* __restrict__ len, int64_t B, int bound, int32_t* __restrict__ state) {
  __shared__ int red[4];
  int m = 0;
  for (int64_t i = threadIdx.x; i < B; i += 256) {
    const int v = len[i];
    m = v > m ? v : m;
    if (v > bound) len[i] = bound;
  }
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    int mm = red[0];
    for (int w = 1; w < 4; ++w) mm = red[w] > mm ? red[w] : mm;
    state[1] = mm;
    state[0] = mm > bound ? 1 : 0;
  }
}
extern "C" int neurst_shrink_clamp(int32_t* new_len, int64_t B, int bound, int32_t* state, void* stream) {
  NEURST_CHECK_ARG(new_len && state && B > 0 && bound > 0, "shrink_clamp: bad arguments");
  shrink_clamp_kernel<<<1, 256, 0, as_stream(stream)>>>(new_len, B, bound, state);
  NEURST_CHECK_LAUNCH("shrink_clamp");
  return NEURST_OK;
}

// ---------------------------------------------------------------- a17 label-smoothed cross entropy
template <typename TG>
__global__ void __launch_bounds__(256) ls_xent_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                     const int32_t* __restrict__ len_labels, float* __restrict__ xent, TG* __restrict__ grad,
                                                     int64_t L, int V, float conf, float grad_scale, int use_cache) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* cache = use_cache ? reinterpret_cast<float*>(smem) : nullptr;
  __shared__ RowStat red[4];
  __shared__ float fred[16];
  const int64_t bl = blockIdx.x, b = bl / L, l = bl % L;
  const float* row = logits + bl * V;
  TG* grow = grad + bl * V;
  const int len = len_labels[b];
  if (l >= len) {
    for (int i = threadIdx.x; i < V; i += blockDim.x) grow[i] = from_f32<TG>(0.f);
    if (threadIdx.x == 0) xent[bl] = 0.f;
    return;
  }
  RowStat st = block_rowstat(row, cache, V, red);
  const float* src = cache ? cache : row;
  float zs = 0.f;
  for (int i = threadIdx.x; i < V; i += blockDim.x) zs += src[i];
  zs = block_sum(zs, fred);
  const float lse = st.m + logf(st.s);
  const float low = (1.0f - conf) / (float)(V - 1);
  const float normalizing = -(conf * logf(conf) + (float)(V - 1) * low * logf(low + 1e-20f));
  int lab = labels[bl];
  lab = lab < 0 ? 0 : (lab >= V ? V - 1 : lab);
  const float inv_len = 1.0f / (float)len;
  if (threadIdx.x == 0) {
    float sum_logp = zs - (float)V * lse;
    float x = -(low * sum_logp + (conf - low) * (src[lab] - lse)) - normalizing;
    xent[bl] = x * inv_len;
  }
  const float sumsoft = conf + (float)(V - 1) * low;
  const float g = grad_scale * inv_len;
  for (int i = threadIdx.x; i < V; i += blockDim.x) {
    float p = expf(src[i] - lse);
    float soft = (i == lab) ? conf : low;
    grow[i] = from_f32<TG>(g * (p * sumsoft - soft));
  }
}
// Register-resident rows (as ctc_frame_reg_kernel): V % 4 == 0, V <= 256 * 4 * NV4, 16-byte aligned rows.  One HBM read
// of the row in 16-byte lanes, maximum first and then one exp per element, one vector store of the gradient
// (the LDS-cached kernel above reads the row with 4-byte lanes and evaluates exp twice: 59 us at 1280 x 10000).
template <typename TG, int NV4>
__global__ void __launch_bounds__(256) ls_xent_reg_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                         const int32_t* __restrict__ len_labels, float* __restrict__ xent, TG* __restrict__ grad,
                                                         int64_t L, int V, float conf, float grad_scale) {
  __shared__ float redm[4], reds[4], redz[4];
  const int64_t bl = blockIdx.x, b = bl / L, l = bl % L;
  const int nv4 = V >> 2, tid = threadIdx.x;
  const f32x4* row4 = reinterpret_cast<const f32x4*>(logits + bl * V);
  TG* grow = grad + bl * V;
  auto store4 = [&](int idx, f32x4 y) {
    if constexpr (sizeof(TG) == 4) *reinterpret_cast<f32x4*>(grow + 4 * idx) = y;
    else *reinterpret_cast<bf16x4*>(grow + 4 * idx) = bf16x4{(bf16_t)y[0], (bf16_t)y[1], (bf16_t)y[2], (bf16_t)y[3]};
  };
  const int len = len_labels[b];
  if (l >= len) {
    for (int idx = tid; idx < nv4; idx += 256) store4(idx, f32x4{0.f, 0.f, 0.f, 0.f});
    if (tid == 0) xent[bl] = 0.f;
    return;
  }
  int lab = labels[bl];
  lab = lab < 0 ? 0 : (lab >= V ? V - 1 : lab);
  f32x4 x[NV4];
  float m = -INFINITY, zs = 0.f;
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int idx = tid + i * 256;
    if (idx < nv4) {
      x[i] = row4[idx];
#pragma unroll
      for (int e = 0; e < 4; ++e) { m = fmaxf(m, x[i][e]); zs += x[i][e]; }
    } else {
      x[i] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
  }
  const float x_lab = logits[bl * V + lab];      // read before any store: the fp32 gradient may overwrite the logits
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  zs = wave_sum(zs);
  if ((tid & 63) == 0) { redm[tid >> 6] = m; redz[tid >> 6] = zs; }
  __syncthreads();
  m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
  zs = redz[0] + redz[1] + redz[2] + redz[3];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { x[i][e] = expf(x[i][e] - m); sum += x[i][e]; }   // exp(-inf) = 0 for the unused tail
  sum = wave_sum(sum);
  if ((tid & 63) == 0) reds[tid >> 6] = sum;
  __syncthreads();                               // also: every thread holds x_lab before the stores below
  sum = reds[0] + reds[1] + reds[2] + reds[3];
  const float lse = m + logf(sum);
  const float low = (1.0f - conf) / (float)(V - 1);
  const float inv_len = 1.0f / (float)len;
  if (tid == 0) {
    const float normalizing = -(conf * logf(conf) + (float)(V - 1) * low * logf(low + 1e-20f));
    const float sum_logp = zs - (float)V * lse;
    xent[bl] = (-(low * sum_logp + (conf - low) * (x_lab - lse)) - normalizing) * inv_len;
  }
  const float sumsoft = conf + (float)(V - 1) * low;
  const float g = grad_scale * inv_len, k = g * sumsoft / sum, glow = g * low, gconf = g * conf;
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int idx = tid + i * 256;
    if (idx < nv4) {
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = x[i][e] * k - ((4 * idx + e == lab) ? gconf : glow);
      store4(idx, y);
    }
  }
}
extern "C" int neurst_ls_xent_fwd_bwd(const float* logits, const int32_t* labels, const int32_t* len_labels, float* xent, void* grad,
                                      int grad_dtype, int64_t B, int64_t L, int64_t V, float confidence, float grad_scale, void* stream) {
  NEURST_CHECK_ARG(logits && labels && len_labels && xent && grad, "ls_xent: null pointer");
  NEURST_CHECK_ARG(B > 0 && L > 0 && V > 1 && confidence > 0.f && confidence <= 1.f, "ls_xent: bad args");
  const int use_cache = V * sizeof(float) <= 64 * 1024;
  NEURST_CHECK_ARG(use_cache || (const void*)grad != (const void*)logits, "ls_xent: in-place gradient needs V*4 <= 64 KiB");
  NEURST_CHECK_ARG(grad_dtype == NEURST_F32 || (const void*)grad != (const void*)logits, "ls_xent: bf16 gradient cannot alias the logits");
  const size_t shm = use_cache ? V * sizeof(float) : 0;
  const bool vec = V % 4 == 0 && ((uintptr_t)logits & 15) == 0 && ((uintptr_t)grad & 15) == 0 && V <= 256 * 4 * 16;
#define LS_XENT_REG(TG, NV4) ls_xent_reg_kernel<TG, NV4><<<(unsigned)(B * L), 256, 0, as_stream(stream)>>>(logits, labels, len_labels, xent, (TG*)grad, L, (int)V, confidence, grad_scale)
  if (vec && grad_dtype == NEURST_BF16) {
    if (V <= 256 * 4 * 3) LS_XENT_REG(bf16_t, 3); else if (V <= 256 * 4 * 10) LS_XENT_REG(bf16_t, 10); else LS_XENT_REG(bf16_t, 16);
  } else if (vec) {
    if (V <= 256 * 4 * 3) LS_XENT_REG(float, 3); else if (V <= 256 * 4 * 10) LS_XENT_REG(float, 10); else LS_XENT_REG(float, 16);
  } else if (grad_dtype == NEURST_BF16)
    ls_xent_kernel<bf16_t><<<(unsigned)(B * L), 256, shm, as_stream(stream)>>>(logits, labels, len_labels, xent, (bf16_t*)grad, L, (int)V, confidence, grad_scale, use_cache);
  else
    ls_xent_kernel<float><<<(unsigned)(B * L), 256, shm, as_stream(stream)>>>(logits, labels, len_labels, xent, (float*)grad, L, (int)V, confidence, grad_scale, use_cache);
  NEURST_CHECK_LAUNCH("ls_xent");
  return NEURST_OK;
}

// ---------------------------------------------------------------- a18 MSE
extern "C" int64_t neurst_mse_blocks(int64_t n) { int64_t g = cdiv64(n, 1024); return g > 1024 ? 1024 : (g < 1 ? 1 : g); }
template <typename T, typename TL>
__global__ void __launch_bounds__(256) mse_kernel(const T* __restrict__ pred, const TL* __restrict__ label, float* __restrict__ partial,
                                                 T* __restrict__ dpred, int64_t n, float gs) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float d = to_f32<T>(pred[i]) - to_f32<TL>(label[i]);
    s += d * d;
    dpred[i] = from_f32<T>(gs * d);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
extern "C" int neurst_mse_fwd_bwd(const void* pred, const void* label, int label_dtype, float* partial, void* dpred, int64_t n,
                                  float grad_scale, int dtype, void* stream) {
  NEURST_CHECK_ARG(pred && label && partial && dpred && n > 0, "mse: bad args");
  int grid = (int)neurst_mse_blocks(n);
  float gs = grad_scale * 2.0f / (float)n;
  hipStream_t s = as_stream(stream);
  if (dtype == NEURST_BF16) {
    if (label_dtype == NEURST_BF16) mse_kernel<bf16_t, bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)pred, (const bf16_t*)label, partial, (bf16_t*)dpred, n, gs);
    else mse_kernel<bf16_t, float><<<grid, 256, 0, s>>>((const bf16_t*)pred, (const float*)label, partial, (bf16_t*)dpred, n, gs);
  } else {
    NEURST_CHECK_ARG(label_dtype == NEURST_F32, "mse: fp32 predictions need fp32 labels");
    mse_kernel<float, float><<<grid, 256, 0, s>>>((const float*)pred, (const float*)label, partial, (float*)dpred, n, gs);
  }
  NEURST_CHECK_LAUNCH("mse");
  return NEURST_OK;
}

// ---------------------------------------------------------------- a2 loss mix
__global__ void loss_combine_kernel(const float* __restrict__ xent, const float* __restrict__ ctc, const float* __restrict__ mse_partial,
                                    int64_t n_partial, int64_t mse_numel, int64_t B, int64_t L, float w_ast, float w_ctc, float w_kd,
                                    float* __restrict__ out4) {
  __shared__ float red[16];
  float mse = 0.f;
  if (mse_partial) {
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n_partial; i += blockDim.x) s += mse_partial[i];
    mse = block_sum(s, red) / (float)mse_numel;
  }
  float s_ast = 0.f, s_ctc = 0.f, s_tot = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += blockDim.x) {
    float ce = 0.f;
    if (xent) for (int64_t l = 0; l < L; ++l) ce += xent[b * L + l];
    float c = ctc ? ctc[b] : 0.f;
    s_ast += ce; s_ctc += c;
    s_tot += w_ctc * c + w_kd * mse + w_ast * ce;
  }
  s_ast = block_sum(s_ast, red);
  s_ctc = block_sum(s_ctc, red);
  s_tot = block_sum(s_tot, red);
  if (threadIdx.x == 0) {
    out4[0] = s_tot / (float)B; out4[1] = s_ast / (float)B; out4[2] = s_ctc / (float)B; out4[3] = mse;
  }
}
extern "C" int neurst_loss_combine(const float* xent, const float* ctc, const float* mse_partial, int64_t n_partial, int64_t mse_numel,
                                   int64_t B, int64_t L, float w_ast, float w_ctc, float w_kd, float* out4, void* stream) {
  NEURST_CHECK_ARG(B > 0 && out4, "loss_combine: bad args");
  loss_combine_kernel<<<1, 256, 0, as_stream(stream)>>>(xent, ctc, mse_partial, n_partial, mse_numel > 0 ? mse_numel : 1, B, L, w_ast, w_ctc, w_kd, out4);
  NEURST_CHECK_LAUNCH("loss_combine");
  return NEURST_OK;
}
