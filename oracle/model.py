"""numpy restatement of `TransformerTPE.build_single_graph` (train loss + hand-derived
backward) and the greedy inference graph.  TEST INFRASTRUCTURE ONLY ("parity
unpinned", see oracle/__init__.py).

Follows st/models/TransformerTPE.py:195-484 (composition and loss mix),
st/models/encoders/transformer_encoder.py:60-153, transformer_TPEencoder.py:63-112,
st/models/decoders/transformer_decoder.py:36-246, fc_decoder.py:13-42,
st/models/kd_layer.py:9-68.  Variable names are the TF scopes those files create.
"""
from dataclasses import dataclass, field

import numpy as np

from . import ops

# dropout call-site ids (shared convention with neurst_amd/st/models/sites.py)
SITE_ENC_IN, SITE_ENC2_IN, SITE_DEC_IN, SITE_KD_ATT = 0, 1, 2, 3
SITE_ENC_BLK, SITE_ENC2_BLK, SITE_DEC_BLK = 100, 300, 500


def blk_site(base, i, j):
    return base + 8 * i + j


@dataclass
class TPEConfig:
    dim_input: int = 480            # data.dim_feature*(num_context+1)   (arguments.py:304-307)
    dim_output: int = 10000         # len(vocab)                          (arguments.py:93)
    feature_channel: int = 1
    enc_blocks: int = 4
    enc_heads: int = 8
    enc_units: int = 768
    enc_att_drop: float = 0.1
    enc_res_drop: float = 0.1
    enc2_blocks: int = 4
    enc2_heads: int = 8
    enc2_units: int = 768
    enc2_att_drop: float = 0.1
    enc2_res_drop: float = 0.1
    dec_blocks: int = 4
    dec_heads: int = 8
    dec_units: int = 768
    dec_att_drop: float = 0.1
    dec_res_drop: float = 0.1
    size_embedding: int = 768
    label_smoothing_confidence: float = 0.9
    ctc_factor: float = 0.5         # model.decoder2.kd_factor
    kd_factor: float = 0.05         # model.kd.kd_factor
    kd_units: int = 768
    kd_heads: int = 4
    kd_att_drop: float = 0.1
    loss_type: str = 'ctc_kd_ce'
    shrink_layer: bool = False
    use_semantic_encoder: bool = True
    use_specAug: bool = False
    use_cnn_resample: bool = False
    cnn_layers: int = 0
    cnn_kernel: tuple = (3, 3)
    cnn_stride: tuple = (2, 2)
    num_filters: int = 0
    sos: int = 2
    eos: int = 3
    max_len: int = 250
    name: str = 'transformer'

    @staticmethod
    def from_yaml_dict(d, dim_output, dim_input=None, sos=2, eos=3):
        m = d['model']
        data = d.get('data', {})
        g = lambda dd, k, default=None: dd.get(k, default) if dd else default
        dim_feature = g(data, 'dim_raw_input', 80)
        din = dim_input or dim_feature * (g(data, 'num_context', 0) + 1) * (3 if g(data, 'add_delta') else 1)
        kd = m.get('kd') or {}
        cnn = m.get('cnn') or {}
        c = TPEConfig(
            dim_input=din, dim_output=dim_output, feature_channel=g(data, 'feature_channel', 1) or 1,
            enc_blocks=m['encoder']['num_blocks'], enc_heads=m['encoder']['num_heads'],
            enc_units=m['encoder']['num_cell_units'],
            enc_att_drop=m['encoder']['attention_dropout_rate'], enc_res_drop=m['encoder']['residual_dropout_rate'],
            enc2_blocks=m['encoder2']['num_blocks'], enc2_heads=m['encoder2']['num_heads'],
            enc2_units=m['encoder2']['num_cell_units'],
            enc2_att_drop=m['encoder2']['attention_dropout_rate'], enc2_res_drop=m['encoder2']['residual_dropout_rate'],
            dec_blocks=m['decoder']['num_blocks'], dec_heads=m['decoder']['num_heads'],
            dec_units=m['decoder']['num_cell_units'],
            dec_att_drop=m['decoder']['attention_dropout_rate'], dec_res_drop=m['decoder']['residual_dropout_rate'],
            size_embedding=m['decoder']['size_embedding'],
            label_smoothing_confidence=m['decoder']['label_smoothing_confidence'],
            ctc_factor=m['decoder2']['kd_factor'],
            kd_factor=g(kd, 'kd_factor', 0.0), kd_units=g(kd, 'num_cell_units', 0) or 0,
            kd_heads=g(kd, 'num_heads', 1) or 1, kd_att_drop=g(kd, 'attention_dropout_rate', 0.0) or 0.0,
            loss_type=m['loss_type'], shrink_layer=bool(m.get('shrink_layer')),
            use_semantic_encoder=bool(m.get('use_semantic_encoder')), use_specAug=bool(m.get('use_specAug')),
            use_cnn_resample=bool(m.get('use_cnn_resample')),
            cnn_layers=g(cnn, 'layers', 0) or 0,
            cnn_kernel=tuple(int(i) for i in str(g(cnn, 'kernel', '3,3')).split(',')),
            cnn_stride=tuple(int(i) for i in str(g(cnn, 'stride', '2,2')).split(',')),
            num_filters=m.get('num_filters') or 0,
            sos=sos, eos=eos, max_len=d.get('max_len', 250))
        return c

    def zero_dropout(self):
        for k in ('enc_att_drop', 'enc_res_drop', 'enc2_att_drop', 'enc2_res_drop', 'dec_att_drop',
                  'dec_res_drop', 'kd_att_drop'):
            setattr(self, k, 0.0)
        return self


def _glorot(rng, shape, dtype):
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = rf * shape[-2], rf * shape[-1]
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def param_specs(cfg: TPEConfig):
    """Ordered (name, shape, kind) list; kind in {glorot, ones, zeros}.  Names are the
    TF variable scopes (SURVEY.md 8a parameter inventory)."""
    n = cfg.name
    specs = [(f'{n}/embedding', (cfg.dim_output, cfg.size_embedding), 'glorot')]

    def attn_self(prefix, d):
        return [(f'{prefix}/qkv_transform/kernel', (3 * d, d), 'glorot'), (f'{prefix}/qkv_transform/bias', (3 * d,), 'zeros'),
                (f'{prefix}/output_transform/kernel', (d, d), 'glorot'), (f'{prefix}/output_transform/bias', (d,), 'zeros')]

    def attn_cross(prefix, d, dq=None, dm=None):
        return [(f'{prefix}/q_transform/kernel', (d, dq or d), 'glorot'), (f'{prefix}/q_transform/bias', (d,), 'zeros'),
                (f'{prefix}/kv_transform/kernel', (2 * d, dm or d), 'glorot'), (f'{prefix}/kv_transform/bias', (2 * d,), 'zeros'),
                (f'{prefix}/output_transform/kernel', (d, d), 'glorot'), (f'{prefix}/output_transform/bias', (d,), 'zeros')]

    def ln(prefix, idx, d):
        s = 'layer_norm' if idx == 0 else f'layer_norm_{idx}'
        return [(f'{prefix}/{s}/layer_norm_scale', (d,), 'ones'), (f'{prefix}/{s}/layer_norm_bias', (d,), 'zeros')]

    def ffn(prefix, d):
        return [(f'{prefix}/ff_hidden/dense/kernel', (4 * d, d), 'glorot'), (f'{prefix}/ff_hidden/dense/bias', (4 * d,), 'zeros'),
                (f'{prefix}/ff_hidden/dense_1/kernel', (d, 4 * d), 'glorot'), (f'{prefix}/ff_hidden/dense_1/bias', (d,), 'zeros')]

    din = cfg.dim_input
    if cfg.use_cnn_resample:
        cin = cfg.feature_channel
        f = cfg.dim_input // cfg.feature_channel
        for i in range(cfg.cnn_layers):
            p = f'{n}/encoder/conv_{i}'
            specs += [(f'{p}/conv/kernel', (cfg.cnn_kernel[0], cfg.cnn_kernel[1], cin, cfg.num_filters), 'glorot'),
                      (f'{p}/conv/bias', (cfg.num_filters,), 'zeros')]
            specs += ln(p, 0, cfg.num_filters)
            cin = cfg.num_filters
            f = -(-f // cfg.cnn_stride[1])
        din = f * cfg.num_filters
    specs.append((f'{n}/encoder/encoder_fc/kernel', (din, cfg.enc_units), 'glorot'))
    for i in range(cfg.enc_blocks):
        p = f'{n}/encoder/block_{i}'
        specs += attn_self(f'{p}/encoder_self_attention', cfg.enc_units) + ln(p, 0, cfg.enc_units)
        specs += ffn(p, cfg.enc_units) + ln(p, 1, cfg.enc_units)
    specs.append((f'{n}/decoder/decoder/fully_connected/kernel', (cfg.enc_units, cfg.dim_output), 'glorot'))
    if cfg.use_semantic_encoder:
        for i in range(cfg.enc2_blocks):
            p = f'{n}/TPEncoder/block_{i}'
            specs += attn_self(f'{p}/encoder_self_attention', cfg.enc2_units) + ln(p, 0, cfg.enc2_units)
            specs += ffn(p, cfg.enc2_units) + ln(p, 1, cfg.enc2_units)
    for i in range(cfg.dec_blocks):
        p = f'{n}/decoder/block_{i}'
        d = cfg.dec_units
        specs += attn_self(f'{p}/decoder_self_attention', d) + ln(p, 0, d)
        specs += attn_cross(f'{p}/decoder_vanilla_attention', d, dm=cfg.enc2_units) + ln(p, 1, d)
        specs += ffn(p, d) + ln(p, 2, d)
    specs.append((f'{n}/decoder/decoder_fc/kernel', (cfg.dec_units, cfg.dim_output), 'glorot'))
    if 'kd' in cfg.loss_type:
        specs += attn_cross(f'{n}/decoder_vanilla_attention', cfg.kd_units, dq=cfg.kd_units, dm=cfg.enc2_units)
    return specs


def init_params(cfg, seed=1, dtype=np.float32, perturb=False):
    """Glorot-uniform matrices (TF default for get_variable), LN scale 1, biases 0.
    perturb=True randomises the 1-D parameters too so parity tests exercise them."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, shape, kind in param_specs(cfg):
        if kind == 'glorot':
            params[name] = _glorot(rng, shape, dtype)
        elif perturb:
            base = 1.0 if kind == 'ones' else 0.0
            params[name] = (base + 0.1 * rng.standard_normal(shape)).astype(dtype)
        else:
            params[name] = (np.ones if kind == 'ones' else np.zeros)(shape, dtype)
    return params


class OracleTPE:
    def __init__(self, cfg: TPEConfig, is_train=True, seed=0, step=0, dropout=True):
        self.cfg = cfg
        self.is_train = is_train
        self.seed, self.step = seed, step
        self.dropout = dropout and is_train

    # ---- helpers --------------------------------------------------------
    def _rate(self, r):
        return float(r) if self.dropout else 0.0

    def _key(self, site):
        return ops.site_key(self.seed, site, self.step)

    def _self_block_fwd(self, P, p, aname, x, H, mask_mode, len_k, att_r, res_r, site0):
        d = x.shape[-1]
        qkv = ops.dense_fwd(x, P[f'{p}/{aname}/qkv_transform/kernel'], P[f'{p}/{aname}/qkv_transform/bias'])
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        a, acache = ops.mha_core_fwd(q, k, v, H, mask_mode, len_k, att_r, self._key(site0))
        o = ops.dense_fwd(a, P[f'{p}/{aname}/output_transform/kernel'], P[f'{p}/{aname}/output_transform/bias'])
        h, rcache = ops.residual_fwd(x, o, P[f'{p}/layer_norm/layer_norm_scale'], P[f'{p}/layer_norm/layer_norm_bias'],
                                     res_r, self._key(site0 + 1))
        return h, (x, a, acache, rcache)

    def _self_block_bwd(self, P, G, p, aname, dh, cache):
        x, a, acache, rcache = cache
        d = x.shape[-1]
        dz, do, G[f'{p}/layer_norm/layer_norm_scale'], G[f'{p}/layer_norm/layer_norm_bias'] = \
            ops.residual_bwd(dh, rcache, P[f'{p}/layer_norm/layer_norm_scale'])
        da, G[f'{p}/{aname}/output_transform/kernel'], G[f'{p}/{aname}/output_transform/bias'] = \
            ops.dense_bwd(do, a, P[f'{p}/{aname}/output_transform/kernel'])
        dq, dk, dv = ops.mha_core_bwd(da, acache)
        dqkv = np.concatenate([dq, dk, dv], -1)
        dx, G[f'{p}/{aname}/qkv_transform/kernel'], G[f'{p}/{aname}/qkv_transform/bias'] = \
            ops.dense_bwd(dqkv, x, P[f'{p}/{aname}/qkv_transform/kernel'])
        return dz + dx

    def _cross_fwd(self, P, pa, xq, mem, H, mask_mode, len_k, att_r, key):
        d = P[f'{pa}/q_transform/kernel'].shape[0]
        q = ops.dense_fwd(xq, P[f'{pa}/q_transform/kernel'], P[f'{pa}/q_transform/bias'])
        kv = ops.dense_fwd(mem, P[f'{pa}/kv_transform/kernel'], P[f'{pa}/kv_transform/bias'])
        a, acache = ops.mha_core_fwd(q, kv[..., :d], kv[..., d:], H, mask_mode, len_k, att_r, key)
        o = ops.dense_fwd(a, P[f'{pa}/output_transform/kernel'], P[f'{pa}/output_transform/bias'])
        return o, (xq, mem, a, acache)

    def _cross_bwd(self, P, G, pa, do, cache):
        xq, mem, a, acache = cache
        da, G[f'{pa}/output_transform/kernel'], G[f'{pa}/output_transform/bias'] = \
            ops.dense_bwd(do, a, P[f'{pa}/output_transform/kernel'])
        dq, dk, dv = ops.mha_core_bwd(da, acache)
        dxq, G[f'{pa}/q_transform/kernel'], G[f'{pa}/q_transform/bias'] = ops.dense_bwd(dq, xq, P[f'{pa}/q_transform/kernel'])
        dmem, G[f'{pa}/kv_transform/kernel'], G[f'{pa}/kv_transform/bias'] = \
            ops.dense_bwd(np.concatenate([dk, dv], -1), mem, P[f'{pa}/kv_transform/kernel'])
        return dxq, dmem

    def _ffn_fwd(self, P, p, ln_idx, x, res_r, site):
        w1, b1 = P[f'{p}/ff_hidden/dense/kernel'], P[f'{p}/ff_hidden/dense/bias']
        w2, b2 = P[f'{p}/ff_hidden/dense_1/kernel'], P[f'{p}/ff_hidden/dense_1/bias']
        f1 = ops.dense_fwd(x, w1, b1, relu=True)
        f2 = ops.dense_fwd(f1, w2, b2)
        s = 'layer_norm' if ln_idx == 0 else f'layer_norm_{ln_idx}'
        h, rcache = ops.residual_fwd(x, f2, P[f'{p}/{s}/layer_norm_scale'], P[f'{p}/{s}/layer_norm_bias'],
                                     res_r, self._key(site))
        return h, (x, f1, rcache, s)

    def _ffn_bwd(self, P, G, p, dh, cache):
        x, f1, rcache, s = cache
        dz, df2, G[f'{p}/{s}/layer_norm_scale'], G[f'{p}/{s}/layer_norm_bias'] = \
            ops.residual_bwd(dh, rcache, P[f'{p}/{s}/layer_norm_scale'])
        df1, G[f'{p}/ff_hidden/dense_1/kernel'], G[f'{p}/ff_hidden/dense_1/bias'] = \
            ops.dense_bwd(df2, f1, P[f'{p}/ff_hidden/dense_1/kernel'])
        dx, G[f'{p}/ff_hidden/dense/kernel'], G[f'{p}/ff_hidden/dense/bias'] = \
            ops.dense_bwd(df1, x, P[f'{p}/ff_hidden/dense/kernel'], y=f1, relu=True)
        return dz + dx

    # ---- encoders (transformer_encoder.py:60-153, transformer_TPEencoder.py:63-112)
    def _encoder_fwd(self, P, scope, x, lens, nblk, H, att_r, res_r, site_in, site_blk, use_fc):
        cfg = self.cfg
        caches = {}
        if use_fc and cfg.use_cnn_resample:
            B, T, _ = x.shape
            f = cfg.dim_input // cfg.feature_channel
            cur = x.reshape(B, T, f, cfg.feature_channel)
            conv_caches = []
            for i in range(cfg.cnn_layers):
                p = f'{scope}/conv_{i}'
                c, ccache = ops.conv2d_fwd(cur, P[f'{p}/conv/kernel'], P[f'{p}/conv/bias'], cfg.cnn_stride)
                l, lcache = ops.layer_norm_fwd(c, P[f'{p}/layer_norm/layer_norm_scale'], P[f'{p}/layer_norm/layer_norm_bias'])
                r = np.maximum(l, 0)
                conv_caches.append((cur.shape, ccache, lcache, l))
                cur = r
                lens = np.ceil(lens.astype(np.float32) / cfg.cnn_stride[0]).astype(np.int32)
            pooled, parg = ops.maxpool3x3_fwd(cur)
            caches['cnn'] = (conv_caches, parg, cur.shape)
            x = pooled.reshape(pooled.shape[0], pooled.shape[1], -1)
        B, T, C = x.shape
        sig = ops.timing_signal(T, C, x.dtype)
        h, m_in = ops.dropout_fwd(x + sig, self._rate(res_r), self._key(site_in))
        caches['m_in'] = m_in
        if use_fc:
            caches['fc_in'] = h
            h = ops.tfdense_fwd(h, P[f'{scope}/encoder_fc/kernel'])
        blocks = []
        for i in range(nblk):
            p = f'{scope}/block_{i}'
            h, c1 = self._self_block_fwd(P, p, 'encoder_self_attention', h, H, ops.MASK_KEYPAD, lens,
                                         self._rate(att_r), self._rate(res_r), blk_site(site_blk, i, 0))
            h, c2 = self._ffn_fwd(P, p, 1, h, self._rate(res_r), blk_site(site_blk, i, 2))
            blocks.append((c1, c2))
        keep = (np.arange(T)[None, :] < lens[:, None]).astype(h.dtype)[..., None]
        caches['blocks'], caches['keep'] = blocks, keep
        return h * keep, lens, caches

    def _encoder_bwd(self, P, G, scope, dout, caches, nblk, use_fc):
        cfg = self.cfg
        dh = dout * caches['keep']
        for i in reversed(range(nblk)):
            p = f'{scope}/block_{i}'
            c1, c2 = caches['blocks'][i]
            dh = self._ffn_bwd(P, G, p, dh, c2)
            dh = self._self_block_bwd(P, G, p, 'encoder_self_attention', dh, c1)
        if use_fc:
            dh, G[f'{scope}/encoder_fc/kernel'] = ops.tfdense_bwd(dh, caches['fc_in'], P[f'{scope}/encoder_fc/kernel'])
        if caches['m_in'] is not None:
            dh = dh * caches['m_in']
        if use_fc and cfg.use_cnn_resample:
            conv_caches, parg, pshape = caches['cnn']
            d = ops.maxpool3x3_bwd(dh.reshape(pshape), parg)
            for i in reversed(range(cfg.cnn_layers)):
                p = f'{scope}/conv_{i}'
                xshape, ccache, lcache, l = conv_caches[i]
                d = d * (l > 0)
                d, G[f'{p}/layer_norm/layer_norm_scale'], G[f'{p}/layer_norm/layer_norm_bias'] = \
                    ops.layer_norm_bwd(d, lcache, P[f'{p}/layer_norm/layer_norm_scale'])
                d, G[f'{p}/conv/kernel'], G[f'{p}/conv/bias'] = ops.conv2d_bwd(d, ccache, P[f'{p}/conv/kernel'], cfg.cnn_stride, xshape)
            return None
        return dh  # gradient wrt encoder input (needed by the semantic encoder <- acoustic path)

    # ---- decoder (transformer_decoder.py:184-246) ------------------------
    def _decoder_fwd(self, P, dec_in, enc, len_enc):
        cfg = self.cfg
        n = cfg.name
        emb = P[f'{n}/embedding'][dec_in]                       # decoder.py:198-204
        L = dec_in.shape[1]
        x = emb + (emb + ops.timing_signal(L, emb.shape[-1], emb.dtype))   # transformer_decoder.py:193-195
        h, m_in = ops.dropout_fwd(x, self._rate(cfg.dec_res_drop), self._key(SITE_DEC_IN))
        blocks = []
        for i in range(cfg.dec_blocks):
            p = f'{n}/decoder/block_{i}'
            h, c1 = self._self_block_fwd(P, p, 'decoder_self_attention', h, cfg.dec_heads, ops.MASK_CAUSAL, None,
                                         self._rate(cfg.dec_att_drop), self._rate(cfg.dec_res_drop),
                                         blk_site(SITE_DEC_BLK, i, 0))
            o, c2 = self._cross_fwd(P, f'{p}/decoder_vanilla_attention', h, enc, cfg.dec_heads, ops.MASK_KEYPAD, len_enc,
                                    self._rate(cfg.dec_att_drop), self._key(blk_site(SITE_DEC_BLK, i, 2)))
            h2, r2 = ops.residual_fwd(h, o, P[f'{p}/layer_norm_1/layer_norm_scale'], P[f'{p}/layer_norm_1/layer_norm_bias'],
                                      self._rate(cfg.dec_res_drop), self._key(blk_site(SITE_DEC_BLK, i, 3)))
            h, c3 = self._ffn_fwd(P, p, 2, h2, self._rate(cfg.dec_res_drop), blk_site(SITE_DEC_BLK, i, 4))
            blocks.append((c1, c2, r2, c3))
        logits = ops.tfdense_fwd(h, P[f'{n}/decoder/decoder_fc/kernel'])
        return logits, (dec_in, m_in, blocks, h)

    def _decoder_bwd(self, P, G, dlogits, cache, enc_shape):
        cfg = self.cfg
        n = cfg.name
        dec_in, m_in, blocks, hlast = cache
        dh, G[f'{n}/decoder/decoder_fc/kernel'] = ops.tfdense_bwd(dlogits, hlast, P[f'{n}/decoder/decoder_fc/kernel'])
        denc = np.zeros(enc_shape, dlogits.dtype)
        for i in reversed(range(cfg.dec_blocks)):
            p = f'{n}/decoder/block_{i}'
            c1, c2, r2, c3 = blocks[i]
            dh = self._ffn_bwd(P, G, p, dh, c3)
            dz, do, G[f'{p}/layer_norm_1/layer_norm_scale'], G[f'{p}/layer_norm_1/layer_norm_bias'] = \
                ops.residual_bwd(dh, r2, P[f'{p}/layer_norm_1/layer_norm_scale'])
            dxq, dmem = self._cross_bwd(P, G, f'{p}/decoder_vanilla_attention', do, c2)
            denc += dmem
            dh = self._self_block_bwd(P, G, p, 'decoder_self_attention', dz + dxq, c1)
        if m_in is not None:
            dh = dh * m_in
        demb = np.zeros_like(P[f'{n}/embedding'])
        np.add.at(demb, dec_in.reshape(-1), 2.0 * dh.reshape(-1, dh.shape[-1]))
        G[f'{n}/embedding'] = demb
        return denc

    # ---- full train graph -------------------------------------------------
    def forward(self, P, batch, specaug_params=None):
        cfg = self.cfg
        n = cfg.name
        feat, len_fea = batch['feature'], np.asarray(batch['len_fea'], np.int32)
        st = {}
        if self.is_train and cfg.use_specAug and specaug_params is not None:   # TransformerTPE.py:219-228
            feat = ops.specaug_apply(feat, specaug_params)
        enc_am, len_am, st['enc'] = self._encoder_fwd(
            P, f'{n}/encoder', feat, len_fea, cfg.enc_blocks, cfg.enc_heads, cfg.enc_att_drop, cfg.enc_res_drop,
            SITE_ENC_IN, SITE_ENC_BLK, use_fc=True)
        st['enc_am'], st['len_am'] = enc_am, len_am
        logits_am = ops.tfdense_fwd(enc_am, P[f'{n}/decoder/decoder/fully_connected/kernel'])   # fc_decoder.py:33-38
        st['logits_am'] = logits_am
        st['preds_am'], _ = ops.ctc_greedy_decode(logits_am, len_am)
        x2, len2 = enc_am, len_am
        if cfg.shrink_layer:                                                    # TransformerTPE.py:262-268
            x2, len2, st['shrink_flag'] = ops.shrink_fwd(enc_am, len_am, logits_am)
        if cfg.use_semantic_encoder:
            enc, len_enc, st['enc2'] = self._encoder_fwd(
                P, f'{n}/TPEncoder', x2, len2, cfg.enc2_blocks, cfg.enc2_heads, cfg.enc2_att_drop, cfg.enc2_res_drop,
                SITE_ENC2_IN, SITE_ENC2_BLK, use_fc=False)
        else:
            enc, len_enc = x2, len2
        st['enc_out'], st['len_enc'] = enc, len_enc
        return st

    def loss_and_grads(self, P, batch, specaug_params=None):
        cfg = self.cfg
        n = cfg.name
        st = self.forward(P, batch, specaug_params)
        enc, len_enc = st['enc_out'], st['len_enc']
        B = enc.shape[0]
        dt = enc.dtype.type
        labels, len_labels = batch['label'], np.asarray(batch['len_label'], np.int32)
        dec_in = ops.build_decoder_input(labels, len_labels, cfg.sos)
        logits, dcache = self._decoder_fwd(P, dec_in, enc, len_enc)
        G = {}
        loss_ast, dlogits = ops.ls_xent_fwd_bwd(logits, labels, len_labels, cfg.label_smoothing_confidence)
        loss_ctc = np.zeros(B, enc.dtype)
        loss_bert = dt(0)
        dlogits_am = None
        kd_cache = None
        lt = cfg.loss_type
        if lt == 'CE':
            w_ast, w_ctc, w_kd = 1.0, 0.0, 0.0
        elif lt == 'ce_ctc':                                                    # TransformerTPE.py:385-398
            w_ctc, w_kd = cfg.ctc_factor, 0.0
            w_ast = 1 - cfg.ctc_factor
        elif lt == 'ctc_kd_ce':                                                 # TransformerTPE.py:400-418
            w_ctc, w_kd = cfg.ctc_factor, cfg.kd_factor
            w_ast = 1 - cfg.ctc_factor - cfg.kd_factor
        else:
            raise NotImplementedError(lt)
        if w_ctc:
            loss_ctc, dlogits_am = ops.ctc_loss_fwd_bwd(st['logits_am'], st['len_am'], batch['aux_label'],
                                                        np.asarray(batch['len_aux_label'], np.int32))
        denc = np.zeros_like(enc)
        if w_kd:
            bert = batch['bert_feat']
            pa = f'{n}/decoder_vanilla_attention'
            # kd_layer.py:33-47: dropout uses kd.attention_dropout_rate regardless of is_train
            kd_rate = cfg.kd_att_drop if self.dropout else 0.0
            feat, kd_cache = self._cross_fwd(P, pa, bert, enc, cfg.kd_heads, ops.MASK_NONE, None, kd_rate,
                                             self._key(SITE_KD_ATT))
            loss_bert, dfeat = ops.mse_fwd_bwd(bert, feat)
            _, dmem = self._cross_bwd(P, G, pa, dfeat * dt(w_kd), kd_cache)
            denc += dmem
        loss_vec = dt(w_ctc) * loss_ctc + dt(w_kd) * loss_bert + dt(w_ast) * loss_ast
        loss = loss_vec.mean()
        # backward: d loss / d loss_b = 1/B
        denc += self._decoder_bwd(P, G, dlogits * dt(w_ast / B), dcache, enc.shape)
        if cfg.use_semantic_encoder:
            dx2 = self._encoder_bwd(P, G, f'{n}/TPEncoder', denc, st['enc2'], cfg.enc2_blocks, use_fc=False)
        else:
            dx2 = denc
        T = st['enc_am'].shape[1]
        d_am = ops.shrink_bwd(dx2, st['shrink_flag'], T) if cfg.shrink_layer else dx2
        wk = f'{n}/decoder/decoder/fully_connected/kernel'
        if w_ctc:
            d2, G[wk] = ops.tfdense_bwd(dlogits_am * dt(w_ctc / B), st['enc_am'], P[wk])
            d_am = d_am + d2
        else:
            G[wk] = np.zeros_like(P[wk])
        self._encoder_bwd(P, G, f'{n}/encoder', d_am, st['enc'], cfg.enc_blocks, use_fc=True)
        return {'loss': loss, 'loss_ast': loss_ast.mean(), 'loss_ctc': loss_ctc.mean(), 'loss_bert': loss_bert,
                'logits': logits, 'logits_am': st['logits_am'], 'enc_am': st['enc_am'], 'enc_out': enc,
                'len_enc': len_enc, 'preds_am': st['preds_am'], 'grads': G}

    # ---- greedy inference (transformer_decoder.py:52-182, TransformerTPE.py:486-513)
    def _cached_step(self, P, preds, cache, enc, len_enc):
        """decoder_with_caching_impl + decoder_fc on the last position (transformer_decoder.py:123-182, :65-75):
        re-embeds the whole prefix, runs every block on the last position only, caches block outputs."""
        cfg = self.cfg
        n = cfg.name
        nb, d = cfg.dec_blocks, cfg.dec_units
        emb = P[f'{n}/embedding'][preds]
        x = emb + (emb + ops.timing_signal(preds.shape[1], emb.shape[-1], emb.dtype))
        new_cache = []
        for bi in range(nb):
            p = f'{n}/decoder/block_{bi}'
            pa = f'{p}/decoder_self_attention'
            qkv = ops.dense_fwd(x, P[f'{pa}/qkv_transform/kernel'], P[f'{pa}/qkv_transform/bias'])
            q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
            a, _ = ops.mha_core_fwd(q[:, -1:], k, v, cfg.dec_heads, ops.MASK_NONE)
            o = ops.dense_fwd(a, P[f'{pa}/output_transform/kernel'], P[f'{pa}/output_transform/bias'])
            h, _ = ops.residual_fwd(x[:, -1:], o, P[f'{p}/layer_norm/layer_norm_scale'], P[f'{p}/layer_norm/layer_norm_bias'])
            o, _ = self._cross_fwd(P, f'{p}/decoder_vanilla_attention', h, enc, cfg.dec_heads, ops.MASK_KEYPAD, len_enc, 0.0, 0)
            h, _ = ops.residual_fwd(h, o, P[f'{p}/layer_norm_1/layer_norm_scale'], P[f'{p}/layer_norm_1/layer_norm_bias'])
            h, _ = self._ffn_fwd(P, p, 2, h, 0.0, 0)
            x = np.concatenate([cache[:, :, bi, :], h], 1)
            new_cache.append(x[:, :, None, :])
        cache = np.concatenate(new_cache, 2)
        return ops.tfdense_fwd(x[:, -1, :], P[f'{n}/decoder/decoder_fc/kernel']), cache

    def infer_beam(self, P, features, len_feas, beam_size, length_penalty_weight=0.0):
        """build_infer_graph with beam_size > 1 (TransformerTPE.py:314-319): best-beam ids [B, steps] (not masked past
        <eos>, as the reference returns them), CTC greedy preds, and the full beam_decode_rerank result."""
        assert not self.is_train
        st = self.forward(P, {'feature': features, 'len_fea': len_feas})
        res = beam_decode_rerank(self, P, st['enc_out'], st['len_enc'], beam_size, length_penalty_weight)
        return res[2], st['preds_am'], res

    def infer(self, P, features, len_feas):
        assert not self.is_train
        cfg = self.cfg
        n = cfg.name
        st = self.forward(P, {'feature': features, 'len_fea': len_feas})
        enc, len_enc = st['enc_out'], st['len_enc']
        B, T, _ = enc.shape
        preds = np.full((B, 1), cfg.sos, np.int32)
        finished = np.zeros(B, bool)
        len_decoded = np.ones(B, np.int32)
        nb, d = cfg.dec_blocks, cfg.dec_units
        cache = np.zeros((B, 0, nb, d), enc.dtype)
        i = 0
        max_steps = min(T + 50, cfg.max_len)
        while (not finished.all()) and i < max_steps:
            cur_logit, cache = self._cached_step(P, preds, cache, enc, len_enc)
            cur_ids = cur_logit.argmax(-1).astype(np.int32)
            preds = np.concatenate([preds, cur_ids[:, None]], 1)
            finished = finished | (cur_ids == cfg.eos)
            len_decoded = len_decoded + (1 - finished.astype(np.int32))
            i += 1
        len_decoded = len_decoded - (1 - finished.astype(np.int32))
        preds = preds[:, 1:]
        not_padding = (np.arange(preds.shape[1])[None, :] < len_decoded[:, None]).astype(np.int32)
        return preds * not_padding, st['preds_am'], len_decoded


def _log_softmax(x):
    m = x.max(-1, keepdims=True)
    return x - (m + np.log(np.exp(x - m).sum(-1, keepdims=True)))


def _top_k(x, k):
    """tf.nn.top_k(sorted=True): descending, equal values by lower index."""
    idx = np.argsort(-x, axis=-1, kind='stable')[..., :k]
    return np.take_along_axis(x, idx, -1), idx.astype(np.int32)


REF_INF = 1e10     # `inf = 1e10`, st/models/decoders/transformer_decoder.py:14


def beam_decode_rerank(model: 'OracleTPE', P, enc, len_enc, beam_size, length_penalty_weight):
    """Transformer_Decoder.beam_decode_rerank (st/models/decoders/transformer_decoder.py:248-431) with lambda_lm = 0 (z_lm
    is zeros there, :309).  Returns (preds_sorted [B, beam, steps], scores_sorted [B, beam], best preds [B, steps],
    len_decoded [B * beam] in the UNSORTED beam order the reference returns, :431)."""
    cfg = model.cfg
    n = cfg.name
    B, T, _ = enc.shape
    K = beam_size
    f32 = np.float32
    enc = np.repeat(enc, K, axis=0)                                   # tile + reshape: rows b * K + j  (:258-262)
    len_enc = np.repeat(len_enc, K)
    R = B * K
    preds = np.full((R, 1), cfg.sos, np.int32)                         # token_init (:265)
    len_decoded = np.ones(R, np.int32)
    scores = np.tile(np.array([0.0] + [-REF_INF] * (K - 1), f32), B)   # (:269-270)
    finished = np.zeros(R, bool)
    nb, d = cfg.dec_blocks, cfg.dec_units
    cache = np.zeros((R, 0, nb, d), enc.dtype)
    base = np.repeat(np.arange(B), K)
    max_steps = min(T + 50, cfg.max_len)
    i = 0
    while (not finished.all()) and i < max_steps:                      # not_finished (:375-383)
        cur_logit, cache = model._cached_step(P, preds, cache, enc, len_enc)
        z = _log_softmax(cur_logit.astype(f32))
        next_scores, next_preds = _top_k(z, K)                         # (:341)
        fin = finished.astype(f32)[:, None]
        next_preds = next_preds * (1 - finished.astype(np.int32)[:, None]) + finished.astype(np.int32)[:, None] * cfg.eos   # get_bias_preds
        b = np.tile(np.array([0.0] + [-REF_INF] * (K - 1), f32)[None, :], (R, 1))
        next_scores = (next_scores * (1 - fin) + b * fin).astype(f32)  # get_bias_scores
        sc = (scores[:, None] + next_scores).astype(f32).reshape(B, K * K)
        ln = (len_decoded[:, None] + (next_preds != cfg.eos).astype(np.int32)).reshape(B, K * K)
        lp = np.power((5 + ln) / (5 + 1), length_penalty_weight)       # float64, as int / int in TF (:353)
        sc = (sc / lp.astype(f32)).astype(f32)
        _, k_idx = _top_k(sc, K)                                       # (:357)
        k_idx = base * K * K + k_idx.reshape(-1)
        len_decoded = ln.reshape(-1)[k_idx]
        scores = sc.reshape(-1)[k_idx]
        nxt = next_preds.reshape(-1)[k_idx]
        parent = k_idx // K
        preds = np.concatenate([preds[parent], nxt[:, None]], 1)
        cache = cache[parent]
        finished = finished | (nxt == cfg.eos)                         # (:368-369) sic: the old flags are NOT re-ordered by parent
        i += 1
    len_decoded = len_decoded - (1 - finished.astype(np.int32))
    preds = preds[:, 1:]
    s_sorted, order = _top_k(scores.reshape(B, K), K)
    order = base * K + order.reshape(-1)
    preds_sorted = preds[order].reshape(B, K, -1)
    return preds_sorted, s_sorted, preds_sorted[:, 0, :], len_decoded


def handle_gradients(grads, P, lambda_l2=0.0, grad_clip_global_norm=0.0):
    """st/tools/tftools/gradientTools.py:40-67 for this model's variables: g += lambda_l2 * v unless 'bias' is in the variable's
    name (:47-48); the value / norm clips (:51-56) only apply to variables named *lstm* - there are none; then
    tf.clip_by_global_norm (:60-65): every g * clip / max(global_norm, clip), global_norm = sqrt(sum of all squares)."""
    out = {}
    for k, g in grads.items():
        out[k] = g + g.dtype.type(lambda_l2) * P[k] if (lambda_l2 > 0 and 'bias' not in k) else g
    if grad_clip_global_norm > 0:
        gn = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in out.values()))
        f = grad_clip_global_norm / max(gn, grad_clip_global_norm)
        out = {k: g * g.dtype.type(f) for k, g in out.items()}
    return out


def train_step(model: OracleTPE, P, M, V, batch, global_step, warmup_steps, peak, decay_steps, specaug_params=None, lambda_l2=0.0,
               grad_clip_global_norm=0.0, frozen_scope=None):
    """One optimisation step: lr from the pre-increment global_step (run_lut.py:346), handle_gradients (TransformerTPE.py:167-168),
    then the frozen-scope filter (TransformerTPE.py:170-183: variables whose name contains one of the comma-separated
    strings are dropped from the list given to apply_gradients - AFTER handle_gradients, so their gradients still count in
    the global norm - and neither they nor their Adam slots change), TF Adam with beta1=.9, beta2=.98, eps=1e-9 hard-coded
    (lstmModel.py:526-530)."""
    model.step = global_step
    out = model.loss_and_grads(P, batch, specaug_params)
    if lambda_l2 or grad_clip_global_norm:
        out['grads'] = handle_gradients(out['grads'], P, lambda_l2, grad_clip_global_norm)
    lr = ops.warmup_exponential_decay(global_step, warmup_steps, peak, 0.5, decay_steps)
    t = global_step + 1
    frozen = [f for f in str(frozen_scope).split(',') if f] if frozen_scope else []
    for k in P:
        if any(f in k for f in frozen):
            continue
        P[k], M[k], V[k] = ops.adam_step(P[k], out['grads'][k], M[k], V[k], lr, t)
    return out, lr
