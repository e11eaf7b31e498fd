#!/usr/bin/env python
"""Regenerates the golden vectors in this directory:  python tests/golden/make_golden.py

PROVENANCE: the reference (TensorFlow 1.15) cannot run in the build container and ships no fixtures for
this path ("parity unpinned"), so these vectors come from the repo's OWN fp64 numpy oracle
(`oracle/`) on fixed numpy seeds - SURVEY.md 8c step (3).  They pin (a) the oracle against silent drift
(tests/test_golden_cpu.py) and (b) the HIP path on the GPU box (tests/test_golden_gpu.py), which never sees
/root/reference.  The only vector here that restates a known answer held by the reference is
`splice.npz:doc_*` (the worked example in the docstring of st/tools/tftools/tfAudioTools.py:69-80, retyped
as data).  Every file stays below ~250 kB.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ops as O  # noqa: E402
from oracle.model import OracleTPE, TPEConfig, init_params  # noqa: E402
from neurst_amd.synthetic import make_batch  # noqa: E402


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('%-22s %7.1f kB' % (name + '.npz', os.path.getsize(path) / 1e3))


def f32(a):
    return np.asarray(a, np.float32)


def gen_splice():
    # known answer held by the reference: docstring example, left_num=0, right_num=2 (typed in as data, not computed)
    doc_in = np.array([[i, i, i] for i in range(1, 8)], np.float32)
    doc_want = np.array([[1, 1, 1, 2, 2, 2, 3, 3, 3], [2, 2, 2, 3, 3, 3, 4, 4, 4], [3, 3, 3, 4, 4, 4, 5, 5, 5], [4, 4, 4, 5, 5, 5, 6, 6, 6],
                         [5, 5, 5, 6, 6, 6, 7, 7, 7], [6, 6, 6, 7, 7, 7, 0, 0, 0], [7, 7, 7, 0, 0, 0, 0, 0, 0]], np.float32)
    rng = np.random.default_rng(100)
    x = f32(rng.standard_normal((3, 20, 4)))
    lens = np.array([20, 13, 1], np.int32)
    x *= (np.arange(20)[None, :, None] < lens[:, None, None])
    out, olen = O.splice_downsample_batch(x, lens, 5, 3)
    save('splice', doc_in=doc_in, doc_want=doc_want, x=x, lens=lens, out=f32(out), out_len=olen)


def gen_layernorm():
    rng = np.random.default_rng(101)
    B, T, C = 2, 12, 16
    x, y = rng.standard_normal((B, T, C)), rng.standard_normal((B, T, C))
    gam, bet = 1 + 0.1 * rng.standard_normal(C), 0.1 * rng.standard_normal(C)
    dout = rng.standard_normal((B, T, C))
    arrs = dict(x=f32(x), y=f32(y), gamma=f32(gam), beta=f32(bet), dout=f32(dout))
    for tag, rate in (('r0', 0.0), ('r25', 0.25)):
        key = O.site_key(7, 100, 0)
        out, cache = O.residual_fwd(arrs['x'].astype(np.float64), arrs['y'].astype(np.float64), arrs['gamma'].astype(np.float64),
                                    arrs['beta'].astype(np.float64), rate, key)
        dz, dy, dg, db = O.residual_bwd(arrs['dout'].astype(np.float64), cache, arrs['gamma'].astype(np.float64))
        arrs.update({tag + '_rate': np.float32(rate), tag + '_key': np.uint32(key), tag + '_out': f32(out), tag + '_dz': f32(dz),
                     tag + '_dy': f32(dy), tag + '_dgamma': f32(dg), tag + '_dbeta': f32(db)})
    save('layernorm', **arrs)


def gen_attention():
    for name, (B, H, Tq, Tk, dh, mode, rate) in {
            'attention_tiny': (2, 2, 12, 12, 8, O.MASK_KEYPAD, 0.0),
            'attention_causal': (2, 2, 12, 12, 8, O.MASK_CAUSAL, 0.0),
            'attention_dh96': (1, 1, 70, 70, 96, O.MASK_KEYPAD, 0.1)}.items():
        rng = np.random.default_rng(102 + dh + mode)
        d = H * dh
        q, k, v, do = (f32(rng.standard_normal((B, T, d))) for T in (Tq, Tk, Tk, Tq))
        lens = np.array([Tk, Tk // 3] if B == 2 else [50], np.int32)
        key = O.site_key(3, 101, 1)
        o, cache = O.mha_core_fwd(q.astype(np.float64), k.astype(np.float64), v.astype(np.float64), H, mode, lens, rate, key)
        dq, dk, dv = O.mha_core_bwd(do.astype(np.float64), cache)
        save(name, q=q, k=k, v=v, do=do, lens=lens, H=np.int32(H), mode=np.int32(mode), rate=np.float32(rate), key=np.uint32(key),
             o=f32(o), dq=f32(dq), dk=f32(dk), dv=f32(dv))


def gen_ctc():
    rng = np.random.default_rng(103)
    B, T, V, U = 5, 12, 11, 4
    logits = f32(rng.standard_normal((B, T, V)) * 2)
    labels = rng.integers(1, V - 1, size=(B, U)).astype(np.int32)
    labels[0, :3] = 5                     # repeated labels need the blank between them
    labels[3, :] = 2                      # all repeats: 4 labels + 3 forced blanks > 6 frames -> no valid path (+inf)
    len_labels = np.array([4, 2, 1, 4, 4], np.int32)
    len_logits = np.array([12, 7, 3, 6, 3], np.int32)   # row 4: U > T -> ignored (loss 0, zero gradient)
    loss, grad = O.ctc_loss_fwd_bwd(logits.astype(np.float64), len_logits, labels, len_labels)
    dense, dl = O.ctc_greedy_decode(logits, len_logits)
    save('ctc', logits=logits, labels=labels, len_labels=len_labels, len_logits=len_logits, loss=loss.astype(np.float64), grad=f32(grad),
         greedy=dense.astype(np.int32), greedy_len=dl.astype(np.int32))


def gen_shrink():
    rng = np.random.default_rng(104)
    B, T, D, V = 3, 12, 16, 11
    logits = f32(rng.standard_normal((B, T, V)))
    logits[..., V - 1] += 1.0
    logits[2, :, V - 1] += 100.0          # nothing kept -> one zero frame, length 0
    x = f32(rng.standard_normal((B, T, D)))
    lens = np.array([12, 7, 12], np.int32)
    out, new_len, flag = O.shrink_fwd(x, lens, logits)
    dout = f32(rng.standard_normal(out.shape)) * (np.arange(out.shape[1])[None, :, None] < new_len[:, None, None])
    save('shrink', x=x, logits=logits, lens=lens, out=f32(out), new_len=new_len.astype(np.int32), dout=f32(dout), dx=f32(O.shrink_bwd(dout, flag, T)))


def gen_xent_mse_adam():
    rng = np.random.default_rng(105)
    B, L, V = 2, 5, 11
    logits = f32(rng.standard_normal((B, L, V)) * 3)
    labels = rng.integers(0, V, size=(B, L)).astype(np.int32)
    lens = np.array([5, 2], np.int32)
    loss, grad = O.ls_xent_fwd_bwd(logits.astype(np.float64), labels, lens, 0.9)
    a, b = f32(rng.standard_normal((2, 5, 16))), f32(rng.standard_normal((2, 5, 16)))
    mse, dmse = O.mse_fwd_bwd(b.astype(np.float64), a.astype(np.float64))
    p, g = f32(rng.standard_normal(300)), f32(rng.standard_normal(300) * 0.1)
    m, v = f32(rng.standard_normal(300) * 0.01), f32(rng.random(300) * 1e-3)
    steps = np.array([1, 7, 12, 500], np.int64)
    lrs = np.array([O.warmup_exponential_decay(s, 10, 1e-3, 0.5, 100) for s in steps], np.float64)
    p2, m2, v2 = O.adam_step(p.astype(np.float64), g.astype(np.float64), m.astype(np.float64), v.astype(np.float64), lrs[1], 8)
    save('xent_mse_adam', logits=logits, labels=labels, lens=lens, xent=loss, dlogits=f32(grad), a=a, b=b, mse=np.float64(mse), da=f32(dmse),
         p=p, g=g, m=m, v=v, t=np.int64(8), lr=np.float64(lrs[1]), p2=f32(p2), m2=f32(m2), v2=f32(v2), lr_steps=steps, lr_values=lrs)


def tiny_cfg(loss_type, shrink):
    d = 16
    blk = dict(num_blocks=2, num_heads=2, num_cell_units=d, attention_dropout_rate=0.0, residual_dropout_rate=0.0)
    y = {'data': {'dim_raw_input': 4, 'num_context': 5, 'downsample': 3, 'add_delta': False, 'feature_channel': 1},
         'model': {'encoder': dict(blk, type='transformer_encoder'), 'encoder2': dict(blk, type='transformer_tpeencoder'),
                   'decoder': dict(blk, type='transformer_decoder', size_embedding=d, label_smoothing_confidence=0.9),
                   'decoder2': {'type': 'fc_decoder', 'ctc_merge_repeated': True, 'kd_factor': 0.5},
                   'kd': {'feat_extract_type': 'self_attention', 'loss_type': 'mean_squared_error', 'kd_factor': 0.05, 'num_cell_units': d,
                          'num_heads': 2, 'attention_dropout_rate': 0.0},
                   'structure': 'transformerTpe', 'loss_type': loss_type, 'use_semantic_encoder': True, 'shrink_layer': shrink},
         'max_len': 12}
    return TPEConfig.from_yaml_dict(y, 11, dim_input=24)   # == __graft_entry__.tiny_args() seen by the oracle


def gen_model():
    for name, loss_type, shrink in (('model_lut', 'ctc_kd_ce', False), ('model_costt', 'ce_ctc', True)):
        cfg = tiny_cfg(loss_type, shrink).zero_dropout()
        rng = np.random.default_rng(106)
        P = init_params(cfg, seed=1, dtype=np.float32)
        for k, v in P.items():
            if v.ndim == 1:               # biases / LN parameters away from their 0 / 1 initial values
                P[k] = (v + 0.1 * rng.standard_normal(v.shape)).astype(np.float32)
        batch = make_batch(B=2, Traw=36, V=11, L=5, U=4, Lb=4, bert_dim=16 if loss_type == 'ctc_kd_ce' else None, dim_raw=4, ragged=True, seed=11)
        feat, lens = O.splice_downsample_batch(batch['feature'], batch['len_fea'], 5, 3)
        P64 = {k: v.astype(np.float64) for k, v in P.items()}
        ref = OracleTPE(cfg, is_train=True, dropout=False).loss_and_grads(P64, dict(batch, feature=feat, len_fea=lens))
        ids, ids_am, dec_len = OracleTPE(cfg, is_train=False).infer(P64, feat, lens)
        arrs = {'param/' + k: v for k, v in P.items()}
        arrs.update({'grad/' + k: f32(v) for k, v in ref['grads'].items()})
        arrs.update({'batch/' + k: v for k, v in batch.items()})
        arrs.update(loss=np.float64(ref['loss']), loss_ast=np.float64(ref['loss_ast']), loss_ctc=np.float64(ref['loss_ctc']),
                    loss_bert=np.float64(ref.get('loss_bert', 0.0)), logits=f32(ref['logits']), logits_am=f32(ref['logits_am']),
                    len_enc=np.asarray(ref['len_enc'], np.int32), greedy=np.asarray(ids, np.int32), greedy_am=np.asarray(ids_am, np.int32),
                    greedy_len=np.asarray(dec_len, np.int32))
        save(name, **arrs)


if __name__ == '__main__':
    gen_splice()
    gen_layernorm()
    gen_attention()
    gen_ctc()
    gen_shrink()
    gen_xent_mse_adam()
    gen_model()
