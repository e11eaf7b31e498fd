"""TEST INFRASTRUCTURE (like the rest of oracle/): an independent torch-CPU (autograd) expression of the TransformerTPE train
graph, used to cross-check the numpy oracle and, in float32, as the second leg of bench.py's cpu_baseline
(the stand-in for "the reference TF graph on CPU" - both bottom out in multithreaded library GEMMs).  It shares no code
with oracle/: attention, layer norm, CTC and the losses go through torch library ops
and autograd instead of the oracle's hand-derived backward.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def timing(length, channels, dtype):
    nts = channels // 2
    inc = math.log(1e4) / (nts - 1)
    inv = torch.exp(torch.arange(nts, dtype=dtype) * -inc)
    st = torch.arange(length, dtype=dtype)[:, None] * inv[None, :]
    sig = torch.cat([torch.sin(st), torch.cos(st)], 1)
    return F.pad(sig, (0, channels % 2))


def mha(q, k, v, H, bias):
    B, Tq, C = q.shape
    dh = C // H
    sp = lambda x: x.view(B, -1, H, dh).transpose(1, 2)
    logits = (sp(q) * dh ** -0.5) @ sp(k).transpose(-1, -2)
    if bias is not None:
        logits = logits + bias
    return (torch.softmax(logits, -1) @ sp(v)).transpose(1, 2).reshape(B, Tq, C)


def keypad_bias(lens, Tk, dtype):
    pad = (torch.arange(Tk)[None, :] >= torch.as_tensor(np.asarray(lens))[:, None]).to(dtype)
    return (pad * -1e9)[:, None, None, :]


def ln(x, P, prefix):
    return F.layer_norm(x, x.shape[-1:], P[prefix + '/layer_norm_scale'], P[prefix + '/layer_norm_bias'], eps=1e-6)


def self_block(P, p, an, x, H, bias):
    d = x.shape[-1]
    qkv = F.linear(x, P[f'{p}/{an}/qkv_transform/kernel'], P[f'{p}/{an}/qkv_transform/bias'])
    a = mha(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, bias)
    return ln(x + F.linear(a, P[f'{p}/{an}/output_transform/kernel'], P[f'{p}/{an}/output_transform/bias']), P, f'{p}/layer_norm')


def cross(P, pa, xq, mem, H, bias):
    d = P[f'{pa}/q_transform/kernel'].shape[0]
    q = F.linear(xq, P[f'{pa}/q_transform/kernel'], P[f'{pa}/q_transform/bias'])
    kv = F.linear(mem, P[f'{pa}/kv_transform/kernel'], P[f'{pa}/kv_transform/bias'])
    a = mha(q, kv[..., :d], kv[..., d:], H, bias)
    return F.linear(a, P[f'{pa}/output_transform/kernel'], P[f'{pa}/output_transform/bias'])


def ffn(P, p, x, lnname):
    f = F.linear(torch.relu(F.linear(x, P[f'{p}/ff_hidden/dense/kernel'], P[f'{p}/ff_hidden/dense/bias'])),
                 P[f'{p}/ff_hidden/dense_1/kernel'], P[f'{p}/ff_hidden/dense_1/bias'])
    return ln(x + f, P, f'{p}/{lnname}')


def encoder(P, cfg, scope, x, lens, nblk, H, use_fc):
    lens = np.asarray(lens)
    if use_fc and cfg.use_cnn_resample:
        B, T, _ = x.shape
        cur = x.view(B, T, cfg.dim_input // cfg.feature_channel, cfg.feature_channel)
        for i in range(cfg.cnn_layers):
            p = f'{scope}/conv_{i}'
            w = P[f'{p}/conv/kernel']                       # [kh,kw,cin,cout]
            kh, kw = w.shape[:2]
            sh, sw = cfg.cnn_stride
            Hh, Ww = cur.shape[1], cur.shape[2]
            oh, ow = -(-Hh // sh), -(-Ww // sw)
            ph, pw = max((oh - 1) * sh + kh - Hh, 0), max((ow - 1) * sw + kw - Ww, 0)
            xin = F.pad(cur.permute(0, 3, 1, 2), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
            c = F.conv2d(xin, w.permute(3, 2, 0, 1), P[f'{p}/conv/bias'], stride=(sh, sw)).permute(0, 2, 3, 1)
            cur = torch.relu(ln(c, P, f'{p}/layer_norm'))
            lens = np.ceil(lens / sh).astype(np.int64)
        cur = F.max_pool2d(cur.permute(0, 3, 1, 2), 3, 1, 1).permute(0, 2, 3, 1)
        x = cur.reshape(cur.shape[0], cur.shape[1], -1)
    B, T, C = x.shape
    h = x + timing(T, C, x.dtype)
    if use_fc:
        h = h @ P[f'{scope}/encoder_fc/kernel']
    bias = keypad_bias(lens, T, x.dtype)
    for i in range(nblk):
        p = f'{scope}/block_{i}'
        h = self_block(P, p, 'encoder_self_attention', h, H, bias)
        h = ffn(P, p, h, 'layer_norm_1')
    keep = (torch.arange(T)[None, :] < torch.as_tensor(lens)[:, None]).to(x.dtype)[..., None]
    return h * keep, lens


def loss_fn(P, cfg, batch, specaug_mask=None):
    """Returns (loss, dict of intermediates).  Dropout must be 0."""
    n = cfg.name
    dt = P[f'{n}/embedding'].dtype
    feat = torch.as_tensor(batch['feature']).to(dt)
    if specaug_mask is not None:
        feat = feat * torch.as_tensor(specaug_mask).to(dt)
    enc_am, len_am = encoder(P, cfg, f'{n}/encoder', feat, batch['len_fea'], cfg.enc_blocks, cfg.enc_heads, True)
    logits_am = enc_am @ P[f'{n}/decoder/decoder/fully_connected/kernel']
    x2, len2 = enc_am, len_am
    if cfg.shrink_layer:
        V = logits_am.shape[-1]
        am = logits_am.argmax(-1)
        prev = torch.cat([torch.full_like(am[:, :1], V - 1), am[:, :-1]], 1)
        flag = (prev != am) & (am != V - 1) & (torch.arange(am.shape[1])[None] < torch.as_tensor(len_am)[:, None])
        len2 = flag.sum(-1).numpy()
        m = max(int(len2.max()), 1)
        rows = []
        for b in range(am.shape[0]):
            r = enc_am[b][flag[b]]
            rows.append(F.pad(r, (0, 0, 0, m - r.shape[0])))
        x2 = torch.stack(rows)
    if cfg.use_semantic_encoder:
        enc, len_enc = encoder(P, cfg, f'{n}/TPEncoder', x2, len2, cfg.enc2_blocks, cfg.enc2_heads, False)
    else:
        enc, len_enc = x2, len2
    labels = torch.as_tensor(batch['label']).long()
    len_labels = torch.as_tensor(np.asarray(batch['len_label'])).long()
    B, L = labels.shape
    mask = (torch.arange(L)[None] < len_labels[:, None])
    dec_in = torch.cat([torch.full((B, 1), cfg.sos), labels[:, :-1]], 1) * mask.long()
    emb = P[f'{n}/embedding'][dec_in]
    h = 2 * emb + timing(L, emb.shape[-1], dt)
    causal = (-1e9 * (1 - torch.tril(torch.ones(L, L, dtype=dt))))[None, None]
    mem_bias = keypad_bias(len_enc, enc.shape[1], dt)
    for i in range(cfg.dec_blocks):
        p = f'{n}/decoder/block_{i}'
        h = self_block(P, p, 'decoder_self_attention', h, cfg.dec_heads, causal)
        h = ln(h + cross(P, f'{p}/decoder_vanilla_attention', h, enc, cfg.dec_heads, mem_bias), P, f'{p}/layer_norm_1')
        h = ffn(P, p, h, 'layer_norm_2')
    logits = h @ P[f'{n}/decoder/decoder_fc/kernel']
    V = logits.shape[-1]
    conf = cfg.label_smoothing_confidence
    low = (1 - conf) / (V - 1)
    normalizing = -(conf * math.log(conf) + (V - 1) * low * math.log(low + 1e-20))
    soft = torch.full_like(logits, low).scatter(-1, labels[..., None], conf)
    xent = -(soft * torch.log_softmax(logits, -1)).sum(-1) - normalizing
    mf = mask.to(dt)
    loss_ast = (xent * mf).sum(-1) / mf.sum(-1)
    loss_ctc = torch.zeros(B, dtype=dt)
    loss_bert = torch.zeros((), dtype=dt)
    if 'ctc' in cfg.loss_type:
        lp = torch.log_softmax(logits_am, -1).transpose(0, 1)
        loss_ctc = F.ctc_loss(lp, torch.as_tensor(batch['aux_label']).long(), torch.as_tensor(np.asarray(len_am)).long(),
                              torch.as_tensor(np.asarray(batch['len_aux_label'])).long(), blank=V - 1, reduction='none',
                              zero_infinity=True)
    if 'kd' in cfg.loss_type:
        bert = torch.as_tensor(batch['bert_feat']).to(dt)
        out = cross(P, f'{n}/decoder_vanilla_attention', bert, enc, cfg.kd_heads, None)
        loss_bert = F.mse_loss(out, bert)
    if cfg.loss_type == 'ctc_kd_ce':
        lv = cfg.ctc_factor * loss_ctc + cfg.kd_factor * loss_bert + (1 - cfg.ctc_factor - cfg.kd_factor) * loss_ast
    elif cfg.loss_type == 'ce_ctc':
        lv = cfg.ctc_factor * loss_ctc + (1 - cfg.ctc_factor) * loss_ast
    else:
        lv = loss_ast
    return lv.mean(), {'logits': logits, 'logits_am': logits_am, 'loss_ast': loss_ast.mean(), 'loss_ctc': loss_ctc.mean(),
                       'loss_bert': loss_bert, 'enc_out': enc}
