"""numpy restatement of every op on the neurst-ST hot path (forward + hand-derived
backward).  TEST INFRASTRUCTURE ONLY - see oracle/__init__.py ("parity unpinned").

All citations are relative to /root/reference/.  Arrays keep the dtype they come in
with (float32 = the reference's arithmetic type, float64 = a tighter yardstick).
Weight layouts follow the reference: `dense` kernels are [out, in]
(st/models/tools/utils.py:388-389), `tf.layers.dense` kernels are [in, out].
"""
import math

import numpy as np

NEG_BIAS = -1e9  # st/layers/common_attention_v1.py:190,203

# --------------------------------------------------------------------------
# counter-based dropout RNG (build-side convention; TF's stream is not
# reproducible).  The HIP kernels use exactly this hash.
# --------------------------------------------------------------------------
_M32 = np.uint64(0xFFFFFFFF)


def fmix32(x):
    """murmur3 finaliser on uint32 values (vectorised, wraps mod 2^32)."""
    x = np.asarray(x, dtype=np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return x


def site_key(seed, site, step=0):
    """32-bit key of one dropout call site in one step."""
    v = (int(seed) * 0x9E3779B9 + int(site) * 0x85EBCA6B + int(step) * 0xC2B2AE35 + 0x27D4EB2F) & 0xFFFFFFFF
    return int(fmix32(np.uint64(v)))


def rate_threshold(rate):
    return int(np.float32(rate) * np.float32(16777216.0))


def keep_mask(key, n, rate):
    """keep[i] for flat element index i in [0, n)."""
    idx = np.arange(n, dtype=np.uint64)
    lo = idx & _M32
    hi = idx >> np.uint64(32)
    h = fmix32(lo ^ fmix32(hi ^ np.uint64(key)))
    return (h >> np.uint64(8)) >= np.uint64(rate_threshold(rate))


# Attention-weight dropout draws its decisions 32 at a time - one 32-bit word per (attention row, 32 consecutive keys), the
# unit the kernels store and exchange (keep bitmap [B*H][Tq][ceil(Tk/32)]) - instead of one 32-bit hash per weight, which
# was 15-20 % of the attention forward.  Word number widx = row * nw + w draws sixteen random words
#     W[k] = xorshift32^(k >> 2)( fmix32(lo(4 widx + (k & 3)) ^ fmix32(hi(4 widx + (k & 3)) ^ key)) ),  k = 0..15
# (four hashed seeds, each followed by its three xorshift32 successors); key 32 w + i of the row owns bit i of every word:
# the 16-bit number U_i with binary digits NOT bit_i(W[k]) (k = 15 most significant) is uniform, and the weight is KEPT iff
# U_i < t16 = round((1 - rate) * 65536), evaluated for all 32 keys at once, digit by digit from the least significant:
#     lt = 0;  for k = 0..15:  lt = (lt | W[k]) if bit k of t16 else (lt & W[k])
# P(keep) = t16 / 65536 (0.899994 for rate 0.1; kept weights are still scaled by 1 / (1 - rate) as tf.nn.dropout does).
def attn_threshold16(rate):
    t = int(np.floor((1.0 - float(np.float32(rate))) * 65536.0 + 0.5))
    return min(max(t, 1), 65535)


def xorshift32(x):
    x = np.asarray(x, dtype=np.uint64) & _M32
    x ^= (x << np.uint64(13)) & _M32
    x ^= x >> np.uint64(17)
    x ^= (x << np.uint64(5)) & _M32
    return x


def attn_keep_words(key, n_rows, nw, rate):
    """uint32 [n_rows, nw]: bit i of word w of a row = keep decision of key 32 w + i (the kernels' keep bitmap)."""
    widx = np.arange(int(n_rows) * int(nw), dtype=np.uint64)
    t16 = attn_threshold16(rate)
    seeds = []
    for j in range(4):
        ctr = widx * np.uint64(4) + np.uint64(j)
        seeds.append(fmix32((ctr & _M32) ^ fmix32((ctr >> np.uint64(32)) ^ np.uint64(key))))
    lt = np.zeros_like(widx)
    chain = list(seeds)
    for k in range(16):
        if k and (k & 3) == 0:
            chain = [xorshift32(c) for c in chain]
        W = chain[k & 3]
        lt = (lt | W) if (t16 >> k) & 1 else (lt & W)
    return lt.astype(np.uint32).reshape(int(n_rows), int(nw))


def attn_keep_mask(key, n_rows, Tk, rate):
    """bool [n_rows, Tk] from attn_keep_words."""
    nw = (int(Tk) + 31) // 32
    words = attn_keep_words(key, n_rows, nw, rate)
    bits = (words[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & np.uint32(1)
    return bits.reshape(int(n_rows), nw * 32)[:, :int(Tk)].astype(bool)


def dropout_fwd(x, rate, key=None, mask=None):
    """tf.nn.dropout(x, keep_prob=1-rate): kept elements scaled by 1/(1-rate).
    Returns (y, m) with m the multiplicative mask (already scaled)."""
    if rate == 0.0 and mask is None:
        return x, None
    if mask is None:
        keep = keep_mask(key, x.size, rate).reshape(x.shape)
        scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))
        mask = keep.astype(x.dtype) * x.dtype.type(scale)
    return x * mask, mask


# --------------------------------------------------------------------------
# a1  splice + down_sample   (st/tools/tftools/tfAudioTools.py:69-98,
#                             st/tools/tftools/tfRecord.py:827-845)
# --------------------------------------------------------------------------
def splice(features, left_num, right_num):
    """tfAudioTools.py:69-88 on one utterance [T, F]."""
    T = features.shape[0]
    pp = np.pad(features, [[left_num, right_num], [0, 0]])
    return np.concatenate([pp[i:i + T] for i in range(left_num + right_num + 1)], axis=1)


def down_sample(features, rate, axis=0):
    """tfAudioTools.py:91-98."""
    idx = np.arange(0, features.shape[axis], rate)
    return np.take(features, idx, axis=axis)


def process_raw_feature(seq_raw_features, num_context, downsample):
    """tfRecord.py:827-845 (add_delta False in both templates)."""
    return down_sample(splice(seq_raw_features, 0, num_context), downsample, axis=0)


def splice_downsample_batch(x, lens, num_context, downsample):
    """Per-utterance process_raw_feature followed by zero padding to the batch max,
    which is what the reader + bucketed batching produce."""
    B, Traw, F = x.shape
    Tout = -(-Traw // downsample)
    out = np.zeros((B, Tout, F * (num_context + 1)), x.dtype)
    new_lens = np.zeros(B, np.int32)
    for b in range(B):
        y = process_raw_feature(x[b, :lens[b]], num_context, downsample)
        out[b, :y.shape[0]] = y
        new_lens[b] = y.shape[0]
    return out, new_lens


# --------------------------------------------------------------------------
# a3  SpecAugment with explicit parameters (tfAudioTools.py:134-212)
# --------------------------------------------------------------------------
def specaug_real_length(inputs):
    """tfAudioTools.py:150-151: frames whose |.|-sum over the feature dim is non-zero."""
    return (np.abs(inputs).sum(-1) != 0).sum(-1).astype(np.int32)


def specaug_apply(inputs, params):
    """inputs [B, T, v]; params [B, 8] int32 = (f, f0, f', f0', t, t0, t', t0')
    following the concat-of-ones/zeros/ones masks at :161-190."""
    out = inputs.copy()
    for b in range(inputs.shape[0]):
        f, f0, f2, f02, t, t0, t2, t02 = [int(v) for v in params[b]]
        out[b, :, f0:f0 + f] = 0
        out[b, :, f02:f02 + f2] = 0
        out[b, t0:t0 + t, :] = 0
        out[b, t02:t02 + t2, :] = 0
    return out


def specaug_sample_params(real_len, v, F, T, p, key):
    """Sampling logic of tfAudioTools.py:159-186 with the build's hash RNG
    (8 draws per utterance, draw j of utterance b uses counter 8*b+j)."""
    B = len(real_len)
    n = B * 8
    idx = np.arange(n, dtype=np.uint64)
    u = fmix32(idx ^ fmix32(np.uint64(key)))

    def randint(r, maxval):  # uniform int in [0, maxval), maxval >= 1
        return int(r % np.uint64(max(int(maxval), 1)))

    params = np.zeros((B, 8), np.int32)
    for b in range(B):
        r = u[8 * b:8 * b + 8]
        tau = int(real_len[b])
        f = randint(r[0], F)
        f0 = randint(r[1], v - f)
        f2 = randint(r[2], F)
        f02 = randint(r[3], v - f2)
        t = randint(r[4], T)
        cap = int(np.float32(p) * np.float32(tau))
        t = min(t, cap)
        t0 = randint(r[5], tau - t)
        prev_t = t
        t2 = randint(r[6], T)
        t2 = max(min(t2 + prev_t, cap) - prev_t, 0)
        t02 = randint(r[7], tau - t2)
        params[b] = (f, f0, f2, f02, t, t0, t2, t02)
    return params


# --------------------------------------------------------------------------
# a5  timing signal (st/layers/common_attention_v1.py:28-70)
# --------------------------------------------------------------------------
def timing_signal(length, channels, dtype=np.float32, min_timescale=1.0, max_timescale=1.0e4):
    dt = np.dtype(dtype).type
    position = np.arange(length).astype(dtype)
    num_timescales = channels // 2
    log_timescale_increment = dt(math.log(float(max_timescale) / float(min_timescale))) / (dt(num_timescales) - dt(1))
    inv_timescales = dt(min_timescale) * np.exp(np.arange(num_timescales).astype(dtype) * -log_timescale_increment)
    scaled_time = position[:, None] * inv_timescales[None, :]
    signal = np.concatenate([np.sin(scaled_time), np.cos(scaled_time)], axis=1)
    signal = np.pad(signal, [[0, 0], [0, channels % 2]])
    return signal.astype(dtype)


# --------------------------------------------------------------------------
# dense layers
# --------------------------------------------------------------------------
def dense_fwd(x, w, b=None, relu=False):
    """st/models/tools/utils.py:367-398: x . w^T (+ b), kernel [units, in]."""
    y = x @ w.T
    if b is not None:
        y = y + b
    if relu:
        y = np.maximum(y, 0)
    return y


def dense_bwd(dy, x, w, y=None, relu=False, has_bias=True):
    if relu:
        dy = dy * (y > 0)
    x2 = x.reshape(-1, x.shape[-1])
    dy2 = dy.reshape(-1, dy.shape[-1])
    dw = dy2.T @ x2
    db = dy2.sum(0) if has_bias else None
    dx = dy @ w
    return dx, dw, db


def tfdense_fwd(x, w):
    """tf.layers.dense(use_bias=False): x . w, kernel [in, units]
    (transformer_encoder.py:117-122, fc_decoder.py:33-38, transformer_decoder.py:42-46)."""
    return x @ w


def tfdense_bwd(dy, x, w):
    x2 = x.reshape(-1, x.shape[-1])
    dy2 = dy.reshape(-1, dy.shape[-1])
    return dy @ w.T, x2.T @ dy2


# --------------------------------------------------------------------------
# a7  layer norm + residual (common_layers_v1.py:342-347, tools/utils.py:324-337)
# --------------------------------------------------------------------------
def layer_norm_fwd(x, scale, bias, epsilon=1e-6):
    mean = x.mean(-1, keepdims=True)
    variance = np.square(x - mean).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(variance + x.dtype.type(epsilon))
    xhat = (x - mean) * rstd
    return xhat * scale + bias, (xhat, rstd)


def layer_norm_bwd(dout, cache, scale):
    xhat, rstd = cache
    dscale = (dout * xhat).reshape(-1, xhat.shape[-1]).sum(0)
    dbias = dout.reshape(-1, xhat.shape[-1]).sum(0)
    g = dout * scale
    dx = rstd * (g - g.mean(-1, keepdims=True) - xhat * (g * xhat).mean(-1, keepdims=True))
    return dx, dscale, dbias


def residual_fwd(inputs, outputs, scale, bias, rate=0.0, key=None, mask=None):
    """LN(inputs + dropout(outputs))  (tools/utils.py:335-337)."""
    o, m = dropout_fwd(outputs, rate, key, mask)
    out, cache = layer_norm_fwd(inputs + o, scale, bias)
    return out, (cache, m)


def residual_bwd(dout, cache, scale):
    ln_cache, m = cache
    dz, dscale, dbias = layer_norm_bwd(dout, ln_cache, scale)
    d_outputs = dz if m is None else dz * m
    return dz, d_outputs, dscale, dbias


# --------------------------------------------------------------------------
# a8  multi-head attention core (common_attention_v1.py:244-266, 325-365;
#     tools/utils.py:545-553)
# --------------------------------------------------------------------------
MASK_NONE, MASK_KEYPAD, MASK_CAUSAL = 0, 1, 2


def split_heads(x, H):
    B, T, C = x.shape
    return x.reshape(B, T, H, C // H).transpose(0, 2, 1, 3)


def combine_heads(x):
    B, H, T, D = x.shape
    return x.transpose(0, 2, 1, 3).reshape(B, T, H * D)


def attention_bias(mask_mode, Tq, Tk, B, len_k, dtype):
    if mask_mode == MASK_NONE:
        return None
    if mask_mode == MASK_KEYPAD:   # common_attention_v1.py:194-204
        pad = (np.arange(Tk)[None, :] >= np.asarray(len_k)[:, None]).astype(dtype)
        return (pad * dtype(NEG_BIAS))[:, None, None, :]
    tri = np.tril(np.ones((Tq, Tk), dtype))   # common_attention_v1.py:180-191
    return (dtype(NEG_BIAS) * (dtype(1.0) - tri))[None, None]


def mha_core_fwd(q, k, v, H, mask_mode, len_k=None, rate=0.0, key=None, drop_mask=None):
    """q [B,Tq,C], k/v [B,Tk,C] (already projected) -> [B,Tq,C]."""
    dt = q.dtype.type
    B, Tq, C = q.shape
    Tk = k.shape[1]
    dh = C // H
    qh = split_heads(q, H) * dt(dh ** -0.5)          # tools/utils.py:548-549
    kh, vh = split_heads(k, H), split_heads(v, H)
    logits = qh @ kh.transpose(0, 1, 3, 2)
    bias = attention_bias(mask_mode, Tq, Tk, B, len_k, dt)
    if bias is not None:
        logits = logits + bias
    logits = logits - logits.max(-1, keepdims=True)
    e = np.exp(logits)
    w = e / e.sum(-1, keepdims=True)
    if drop_mask is None and rate != 0.0:            # tf.nn.dropout on the weights (common_attention_v1.py:357-359), word rule above
        keep = attn_keep_mask(key, B * H * Tq, Tk, rate).reshape(w.shape)
        drop_mask = keep.astype(w.dtype) * dt(np.float32(1.0) / (np.float32(1.0) - np.float32(rate)))
    wd, m = dropout_fwd(w, rate, key, drop_mask)
    out = combine_heads(wd @ vh)
    return out, (qh, kh, vh, w, wd, m, H)


def mha_core_bwd(dout, cache):
    qh, kh, vh, w, wd, m, H = cache
    dt = qh.dtype.type
    dh = qh.shape[-1]
    doh = split_heads(dout, H)
    dvh = wd.transpose(0, 1, 3, 2) @ doh
    dwd = doh @ vh.transpose(0, 1, 3, 2)
    dw = dwd if m is None else dwd * m
    dlogits = w * (dw - (dw * w).sum(-1, keepdims=True))
    dqh = (dlogits @ kh) * dt(dh ** -0.5)
    dkh = dlogits.transpose(0, 1, 3, 2) @ qh
    return combine_heads(dqh), combine_heads(dkh), combine_heads(dvh)


# --------------------------------------------------------------------------
# a11/a12  CTC loss + greedy decode, TF-1.15 semantics
#   call sites: st/models/TransformerTPE.py:74-125, 251-259
#   algorithm: tensorflow/core/util/ctc/ctc_loss_calculator.{h,cc} (TF 1.15, not
#   vendored): blank = num_classes-1, softmax applied internally, labels >= blank
#   terminate the label sequence, ctc_merge_repeated=True transitions,
#   ignore_longer_outputs_than_inputs=True -> rows with U > T get loss 0 / grad 0,
#   no valid path -> loss +inf and gradient = softmax.
# --------------------------------------------------------------------------
def _logsumexp2(a, b):
    m = np.maximum(a, b)
    with np.errstate(invalid='ignore', divide='ignore'):
        r = m + np.log(np.exp(a - m) + np.exp(b - m))
    return np.where(np.isneginf(m), m, r)


def ctc_loss_fwd_bwd(logits, len_logits, labels, len_labels):
    """logits [B,T,V], labels [B,Umax] -> (loss [B], dlogits [B,T,V]) with unit upstream."""
    B, T, V = logits.shape
    dt = logits.dtype
    blank = V - 1
    loss = np.zeros(B, dt)
    grad = np.zeros_like(logits)
    neg_inf = dt.type(-np.inf)
    for b in range(B):
        Tb = int(len_logits[b])
        lab_full = [int(x) for x in labels[b, :int(len_labels[b])]]
        if Tb == 0 or len(lab_full) > Tb:
            continue
        lab = []
        for x in lab_full:
            if x >= blank:
                break
            lab.append(x)
        z = logits[b, :Tb]
        z = z - z.max(-1, keepdims=True)
        e = np.exp(z)
        y = e / e.sum(-1, keepdims=True)
        lp = np.full(2 * len(lab) + 1, blank, np.int64)
        lp[1::2] = lab
        S = lp.shape[0]
        with np.errstate(divide='ignore'):
            logy = np.log(y[:, lp])
        skip = np.zeros(S, bool)
        skip[2:] = (lp[2:] != blank) & (lp[2:] != lp[:-2])
        alpha = np.full((Tb, S), neg_inf, dt)
        alpha[0, 0] = logy[0, 0]
        if S > 1:
            alpha[0, 1] = logy[0, 1]
        for t in range(1, Tb):
            a = alpha[t - 1]
            s = a.copy()
            s[1:] = _logsumexp2(s[1:], a[:-1])
            a2 = np.full(S, neg_inf, dt)
            a2[2:] = np.where(skip[2:], a[:-2], neg_inf)
            s = _logsumexp2(s, a2)
            alpha[t] = s + logy[t]
        beta = np.full((Tb, S), neg_inf, dt)
        beta[Tb - 1, S - 1] = 0
        if S > 1:
            beta[Tb - 1, S - 2] = 0
        for t in range(Tb - 2, -1, -1):
            nb = beta[t + 1] + logy[t + 1]
            s = nb.copy()
            s[:-1] = _logsumexp2(s[:-1], nb[1:])
            n2 = np.full(S, neg_inf, dt)
            n2[:-2] = np.where(skip[2:], nb[2:], neg_inf)
            s = _logsumexp2(s, n2)
            beta[t] = s
        log_p = neg_inf
        for u in range(S):
            log_p = _logsumexp2(log_p, alpha[0, u] + beta[0, u])
        loss[b] = -log_p
        if np.isneginf(log_p):
            grad[b, :Tb] = y
            continue
        with np.errstate(invalid='ignore'):
            post = np.exp(alpha + beta - log_p)
        g = y.copy()
        np.subtract.at(g, (np.arange(Tb)[:, None].repeat(S, 1), lp[None, :].repeat(Tb, 0)), post)
        grad[b, :Tb] = g
    return loss, grad


def ctc_greedy_decode(logits, len_logits):
    """tf.nn.ctc_greedy_decoder(merge_repeated=True) + sparse_to_dense(default 0)
    (TransformerTPE.py:106-125, 251-259).  Returns (dense [B, maxlen] int32, lens)."""
    B, T, V = logits.shape
    blank = V - 1
    am = logits.argmax(-1)
    seqs = []
    for b in range(B):
        prev = blank
        s = []
        for t in range(int(len_logits[b])):
            c = int(am[b, t])
            if c != blank and c != prev:
                s.append(c)
            prev = c
        seqs.append(s)
    maxlen = max([len(s) for s in seqs] + [0])
    dense = np.zeros((B, maxlen), np.int32)
    for b, s in enumerate(seqs):
        dense[b, :len(s)] = s
    return dense, np.array([len(s) for s in seqs], np.int32)


# --------------------------------------------------------------------------
# a13  shrink_layer (st/models/tools/blocks.py:8-38)
# --------------------------------------------------------------------------
def shrink_fwd(encoded, len_encoded, logits):
    B, T, D = encoded.shape
    blank = logits.shape[-1] - 1
    am = logits.argmax(-1)
    prev = np.concatenate([np.full((B, 1), blank, am.dtype), am[:, :-1]], 1)
    flag = (prev != am) & (am != blank) & (np.arange(T)[None, :] < np.asarray(len_encoded)[:, None])
    len_labels = flag.sum(-1).astype(np.int32)
    max_label_len = max(int(len_labels.max()) if B else 0, 1)
    out = np.zeros((B, max_label_len, D), encoded.dtype)
    for b in range(B):
        idx = np.nonzero(flag[b])[0]
        out[b, :len(idx)] = encoded[b, idx]
    return out, len_labels, flag


def shrink_bwd(dout, flag, T):
    B, _, D = dout.shape
    dx = np.zeros((B, T, D), dout.dtype)
    for b in range(B):
        idx = np.nonzero(flag[b])[0]
        dx[b, idx] = dout[b, :len(idx)]
    return dx


# --------------------------------------------------------------------------
# a15  decoder inputs (decoder.py:63-92, tfTools.py:195-198, TransformerTPE.py:335-338)
# --------------------------------------------------------------------------
def build_decoder_input(labels, len_labels, sos):
    inp = np.concatenate([np.full((labels.shape[0], 1), sos, labels.dtype), labels[:, :-1]], 1)
    mask = (np.arange(labels.shape[1])[None, :] < np.asarray(len_labels)[:, None]).astype(labels.dtype)
    return inp * mask


# --------------------------------------------------------------------------
# a17  label-smoothed cross entropy (model_tools.py:17-38, seq2seqModel.py:157-191)
# --------------------------------------------------------------------------
def ls_xent_fwd_bwd(logits, labels, len_labels, confidence):
    """-> (loss [B], dlogits) where dlogits is d(loss_b)/dlogits with unit upstream."""
    dt = logits.dtype.type
    B, L, V = logits.shape
    conf = dt(confidence)
    low = (dt(1.0) - conf) / dt(V - 1)
    normalizing = -(conf * np.log(conf) + dt(V - 1) * low * np.log(low + dt(1e-20)))
    z = logits - logits.max(-1, keepdims=True)
    lse = np.log(np.exp(z).sum(-1, keepdims=True))
    logp = z - lse
    soft = np.full((B, L, V), low, logits.dtype)
    np.put_along_axis(soft, labels[..., None].astype(np.int64), conf, -1)
    xent = -(soft * logp).sum(-1) - normalizing
    mask = (np.arange(L)[None, :] < np.asarray(len_labels)[:, None]).astype(logits.dtype)
    denom = mask.sum(-1)
    loss = (xent * mask).sum(-1) / denom
    p = np.exp(logp)
    # sum(soft) = conf + (V-1)*low, kept explicit (it is 1 up to rounding)
    dlogits = (p * soft.sum(-1, keepdims=True) - soft) * (mask / denom[:, None])[..., None]
    return loss, dlogits


# --------------------------------------------------------------------------
# a18  KD mean-squared error (kd_layer.py:9-11; tf.losses reduction = sum / numel)
# --------------------------------------------------------------------------
def mse_fwd_bwd(labels, predictions):
    d = predictions - labels
    n = d.size
    loss = np.square(d).sum() / d.dtype.type(n)
    return loss, d * d.dtype.type(2.0 / n)


# --------------------------------------------------------------------------
# a4  conv2d subsampler (transformer_encoder.py:18-32, 69-103); TF SAME padding
# --------------------------------------------------------------------------
def same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def conv2d_fwd(x, w, b, stride):
    """x [B,H,W,Cin] NHWC, w [kh,kw,Cin,Cout] (tf.layers.conv2d kernel), SAME."""
    B, H, W, Cin = x.shape
    kh, kw, _, Cout = w.shape
    Ho, pt, pb = same_pad(H, kh, stride[0])
    Wo, pl, pr = same_pad(W, kw, stride[1])
    xp = np.pad(x, [[0, 0], [pt, pb], [pl, pr], [0, 0]])
    out = np.zeros((B, Ho, Wo, Cout), x.dtype)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + (Ho - 1) * stride[0] + 1:stride[0], j:j + (Wo - 1) * stride[1] + 1:stride[1], :]
            out += patch @ w[i, j]
    return out + b, (xp, (pt, pb, pl, pr))


def conv2d_bwd(dout, cache, w, stride, x_shape):
    xp, (pt, pb, pl, pr) = cache
    kh, kw, Cin, Cout = w.shape
    B, Ho, Wo, _ = dout.shape
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    for i in range(kh):
        for j in range(kw):
            sl = (slice(None), slice(i, i + (Ho - 1) * stride[0] + 1, stride[0]),
                  slice(j, j + (Wo - 1) * stride[1] + 1, stride[1]), slice(None))
            patch = xp[sl]
            dw[i, j] = patch.reshape(-1, Cin).T @ dout.reshape(-1, Cout)
            dxp[sl] += dout @ w[i, j].T
    H, W = x_shape[1], x_shape[2]
    dx = dxp[:, pt:pt + H, pl:pl + W, :]
    return dx, dw, dout.reshape(-1, Cout).sum(0)


def maxpool3x3_fwd(x):
    """tf.layers.max_pooling2d(pool 3x3, stride 1, SAME): padding never wins."""
    B, H, W, C = x.shape
    xp = np.pad(x, [[0, 0], [1, 1], [1, 1], [0, 0]], constant_values=-np.inf)
    stack = np.stack([xp[:, i:i + H, j:j + W, :] for i in range(3) for j in range(3)], 0)
    arg = stack.argmax(0)
    return stack.max(0), arg


def maxpool3x3_bwd(dout, arg):
    B, H, W, C = dout.shape
    dxp = np.zeros((B, H + 2, W + 2, C), dout.dtype)
    for k in range(9):
        i, j = divmod(k, 3)
        dxp[:, i:i + H, j:j + W, :] += dout * (arg == k)
    return dxp[:, 1:-1, 1:-1, :]


# --------------------------------------------------------------------------
# a20  LR schedule + TF Adam (model_tools.py:80-87, lstmModel.py:526-530)
# --------------------------------------------------------------------------
def warmup_exponential_decay(global_step, warmup_steps, peak, decay_rate, decay_steps):
    gs = np.float32(global_step)
    ws = np.float32(warmup_steps)
    if gs <= ws:
        return np.float32(peak) * gs / ws
    return np.float32(peak) * np.float32(decay_rate) ** ((gs - ws) / np.float32(decay_steps))


def adam_step(p, g, m, v, lr, t, beta1=0.9, beta2=0.98, eps=1e-9):
    """tf.train.AdamOptimizer update, t = 1 for the first apply_gradients."""
    dt = p.dtype.type
    lr_t = dt(lr) * np.sqrt(dt(1.0) - dt(beta2) ** dt(t)) / (dt(1.0) - dt(beta1) ** dt(t))
    m = dt(beta1) * m + dt(1.0 - beta1) * g
    v = dt(beta2) * v + dt(1.0 - beta2) * g * g
    p = p - lr_t * m / (np.sqrt(v) + dt(eps))
    return p, m, v
