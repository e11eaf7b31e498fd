// Beam search bookkeeping of Transformer_Decoder.beam_decode_rerank (st/models/decoders/transformer_decoder.py:248-431)
// on the device, so that a decoding step needs no read-back:
//   beam_topk   :303-341  z = log_softmax(cur_logit); next_scores, next_preds = top_k(z, beam) (sorted, ties -> lower index)
//   beam_prune  :316-373  finished beams keep ONE live branch that emits <eos> (get_bias_scores / get_bias_preds, inf = 1e10
//                         as the reference defines it, :14), candidate score = beam score + branch score, divided by the
//                         length penalty ((5 + len) / 6) ^ w (computed in fp64 like the reference's int / int -> float64
//                         division, cast to fp32), top_k over the beam^2 candidates of one utterance, and the gathers
//                         of lengths / scores / tokens / parents that follow; `finished` stays with its SLOT and is
//                         or-ed with token == eos (:368-369 do not gather it by parent - kept as the reference has it)
//   gather_rows :361-362  preds / cache rows re-ordered by their parent beam
#include "common.h"

constexpr float REF_INF = 1e10f;   // `inf = 1e10` (transformer_decoder.py:14): finite on purpose

// (value, index) ordering of tf.nn.top_k: larger value first, equal values by lower index
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__device__ __forceinline__ void block_argmax(float& v, int& i, float* sv, int* si) {
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (better(ov, oi, v, i)) { v = ov; i = oi; }
  }
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sv[w] = v; si[w] = i; }
  __syncthreads();
  v = sv[0]; i = si[0];
  for (int k = 1; k < nw; ++k)
    if (better(sv[k], si[k], v, i)) { v = sv[k]; i = si[k]; }
}

// one workgroup per row of logits [R, V]: log-softmax statistics, then `k` rounds of first-maximum arg-max
__global__ void __launch_bounds__(256) beam_topk_kernel(const float* __restrict__ logits, float* __restrict__ out_scores,
                                                        int32_t* __restrict__ out_ids, int64_t V, int k) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ float red[16];
  __shared__ int taken[16];
  const float* row = logits + (int64_t)blockIdx.x * V;
  float m = -INFINITY;
  for (int64_t i = threadIdx.x; i < V; i += blockDim.x) m = fmaxf(m, row[i]);
  m = block_max(m, red);
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < V; i += blockDim.x) s += expf(row[i] - m);
  s = block_sum(s, red);
  const float lse = m + logf(s);
  for (int r = 0; r < k; ++r) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int64_t i = threadIdx.x; i < V; i += blockDim.x) {
      bool used = false;
      for (int t = 0; t < r; ++t) used |= taken[t] == (int)i;
      const float x = row[i];
      if (!used && better(x, (int)i, bv, bi)) { bv = x; bi = (int)i; }
    }
    block_argmax(bv, bi, sv, si);
    if (threadIdx.x == 0) {
      taken[r] = bi;
      out_scores[(int64_t)blockIdx.x * k + r] = bv - lse;
      out_ids[(int64_t)blockIdx.x * k + r] = bi;
    }
    __syncthreads();
  }
}

extern "C" int neurst_beam_topk(const float* logits, float* scores, int32_t* ids, int64_t rows, int64_t V, int k, void* stream) {
  NEURST_CHECK_ARG(logits && scores && ids && rows > 0 && V > 0, "beam_topk: bad args");
  NEURST_CHECK_ARG(k >= 1 && k <= 16 && k <= V, "beam_topk: k must be in 1..16 and <= V");
  beam_topk_kernel<<<(int)rows, 256, 0, as_stream(stream)>>>(logits, scores, ids, V, k);
  NEURST_CHECK_LAUNCH("beam_topk");
  return NEURST_OK;
}

// one wave per utterance: beam^2 <= 64 candidates, lane c = (parent beam c / beam, branch c % beam)
__global__ void __launch_bounds__(64) beam_prune_kernel(const float* __restrict__ next_scores, const int32_t* __restrict__ next_ids,
                                                        const float* __restrict__ scores, const int32_t* __restrict__ len_decoded,
                                                        const int32_t* __restrict__ finished, float* __restrict__ new_scores,
                                                        int32_t* __restrict__ new_len, int32_t* __restrict__ new_finished,
                                                        int32_t* __restrict__ new_tokens, int32_t* __restrict__ parents,
                                                        int32_t* __restrict__ n_unfinished, const int32_t* __restrict__ prev_unfinished,
                                                        int beam, int32_t eos, float lp_weight) {
  const int b = blockIdx.x, c = threadIdx.x, nc = beam * beam;
  if (prev_unfinished && *prev_unfinished == 0) {
    // the reference's while_loop ended after the previous step: a step the host enqueued before it could know that
    // leaves every hypothesis as it is (the host trims the extra <eos> columns)
    if (c < beam) {
      const int j = b * beam + c;
      new_scores[j] = scores[j];
      new_len[j] = len_decoded[j];
      new_finished[j] = finished[j];
      new_tokens[j] = eos;
      parents[j] = j;
    }
    return;
  }
  float sc = -INFINITY;
  int ln = 0, tok = 0;
  if (c < nc) {
    const int j = b * beam + c / beam, mbr = c % beam;
    const bool fin = finished[j] != 0;
    tok = fin ? eos : next_ids[(int64_t)j * beam + mbr];                       // get_bias_preds
    const float ns = fin ? (mbr == 0 ? 0.f : -REF_INF) : next_scores[(int64_t)j * beam + mbr];   // get_bias_scores
    ln = len_decoded[j] + (tok != eos ? 1 : 0);
    const double lp = pow((5.0 + (double)ln) / 6.0, (double)lp_weight);
    sc = (scores[j] + ns) / (float)lp;
  }
  int unfinished = 0;
  for (int r = 0; r < beam; ++r) {
    float bv = sc;
    int bi = c < nc ? c : 0x7fffffff;
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    // every lane now knows the winner; its owner publishes the row and leaves the contest
    const int win_tok = __shfl(tok, bi, 64), win_len = __shfl(ln, bi, 64);
    if (c == 0) {
      const int dst = b * beam + r, parent = b * beam + bi / beam;
      const int f = finished[dst] | (win_tok == eos ? 1 : 0);   // sic (:368-369): the flag of SLOT dst, not of the parent
      new_scores[dst] = bv;
      new_len[dst] = win_len;
      new_tokens[dst] = win_tok;
      parents[dst] = parent;
      new_finished[dst] = f;
      unfinished += f ? 0 : 1;
    }
    if (c == bi) sc = -INFINITY;
  }
  if (c == 0 && unfinished) atomicAdd(n_unfinished, unfinished);
}

extern "C" int neurst_beam_prune(const float* next_scores, const int32_t* next_ids, const float* scores, const int32_t* len_decoded,
                                 const int32_t* finished, float* new_scores, int32_t* new_len, int32_t* new_finished,
                                 int32_t* new_tokens, int32_t* parents, int32_t* n_unfinished, const int32_t* prev_unfinished,
                                 int64_t batch, int beam, int32_t eos, float length_penalty_weight, void* stream) {
  NEURST_CHECK_ARG(next_scores && next_ids && scores && len_decoded && finished && new_scores && new_len && new_finished && new_tokens &&
                       parents && n_unfinished, "beam_prune: null pointer");
  NEURST_CHECK_ARG(batch > 0 && beam >= 1 && beam <= 8, "beam_prune: beam must be in 1..8");
  NEURST_CHECK_ARG(new_scores != scores && new_len != len_decoded && new_finished != finished, "beam_prune: outputs must not alias inputs");
  beam_prune_kernel<<<(int)batch, 64, 0, as_stream(stream)>>>(next_scores, next_ids, scores, len_decoded, finished, new_scores, new_len,
                                                              new_finished, new_tokens, parents, n_unfinished, prev_unfinished, beam, eos,
                                                              length_penalty_weight);
  NEURST_CHECK_LAUNCH("beam_prune");
  return NEURST_OK;
}

// dst[r][0:row_bytes) = src[idx[r]][0:row_bytes) with row strides in bytes (16-byte chunks when everything is aligned)
__global__ void gather_rows_kernel(const char* __restrict__ src, char* __restrict__ dst, const int32_t* __restrict__ idx,
                                   int64_t row_bytes, int64_t src_stride, int64_t dst_stride, int vec) {
  const int64_t r = blockIdx.y;
  const char* s = src + (int64_t)idx[r] * src_stride;
  char* d = dst + r * dst_stride;
  if (vec) {
    const int64_t n = row_bytes >> 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
      reinterpret_cast<f32x4*>(d)[i] = reinterpret_cast<const f32x4*>(s)[i];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_bytes; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
  }
}

extern "C" int neurst_gather_rows(const void* src, void* dst, const int32_t* idx, int64_t rows, int64_t row_bytes, int64_t src_stride,
                                  int64_t dst_stride, void* stream) {
  NEURST_CHECK_ARG(src && dst && idx && rows > 0 && rows < 65536 && row_bytes > 0, "gather_rows: bad args");
  NEURST_CHECK_ARG(src_stride >= row_bytes && dst_stride >= row_bytes && src != dst, "gather_rows: strides smaller than a row, or in place");
  const int vec = (((uintptr_t)src | (uintptr_t)dst | (uintptr_t)row_bytes | (uintptr_t)src_stride | (uintptr_t)dst_stride) & 15) == 0;
  const int64_t units = vec ? row_bytes >> 4 : row_bytes;
  int64_t bx = cdiv64(units, 256);
  if (bx > 64) bx = 64;
  dim3 grid((unsigned)bx, (unsigned)rows);
  gather_rows_kernel<<<grid, 256, 0, as_stream(stream)>>>((const char*)src, (char*)dst, idx, row_bytes, src_stride, dst_stride, vec);
  NEURST_CHECK_LAUNCH("gather_rows");
  return NEURST_OK;
}
