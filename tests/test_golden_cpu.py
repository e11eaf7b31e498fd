"""not-gpu: the numpy oracle reproduces the committed golden vectors (tests/golden/*.npz, made by
tests/golden/make_golden.py), so a silent change of the oracle cannot move the GPU parity target."""
import os

import numpy as np
import pytest

from oracle import ops as O
from oracle.model import OracleTPE
from golden.make_golden import tiny_cfg

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return dict(np.load(os.path.join(G, name + '.npz')))


def close(a, b, tol=1e-6):
    return np.abs(np.asarray(a, np.float64) - b).max() <= tol * max(np.abs(b).max(), 1e-12)


def test_splice_known_answer_and_batch():
    g = load('splice')
    assert np.array_equal(O.splice(g['doc_in'], 0, 2), g['doc_want'])        # the reference's own docstring example
    out, olen = O.splice_downsample_batch(g['x'], g['lens'], 5, 3)
    assert np.array_equal(out, g['out']) and np.array_equal(olen, g['out_len'])


@pytest.mark.parametrize('tag', ['r0', 'r25'])
def test_layernorm(tag):
    g = load('layernorm')
    out, cache = O.residual_fwd(g['x'].astype(np.float64), g['y'].astype(np.float64), g['gamma'].astype(np.float64),
                                g['beta'].astype(np.float64), float(g[tag + '_rate']), int(g[tag + '_key']))
    dz, dy, dg, db = O.residual_bwd(g['dout'].astype(np.float64), cache, g['gamma'].astype(np.float64))
    for got, k in ((out, 'out'), (dz, 'dz'), (dy, 'dy'), (dg, 'dgamma'), (db, 'dbeta')):
        assert close(got, g[tag + '_' + k]), k


@pytest.mark.parametrize('name', ['attention_tiny', 'attention_causal', 'attention_dh96'])
def test_attention(name):
    g = load(name)
    o, cache = O.mha_core_fwd(g['q'].astype(np.float64), g['k'].astype(np.float64), g['v'].astype(np.float64), int(g['H']), int(g['mode']),
                              g['lens'], float(g['rate']), int(g['key']))
    dq, dk, dv = O.mha_core_bwd(g['do'].astype(np.float64), cache)
    assert close(o, g['o']) and close(dq, g['dq']) and close(dk, g['dk']) and close(dv, g['dv'])


def test_ctc_edge_rows():
    g = load('ctc')
    loss, grad = O.ctc_loss_fwd_bwd(g['logits'].astype(np.float64), g['len_logits'], g['labels'], g['len_labels'])
    assert np.isposinf(g['loss'][3]) and g['loss'][4] == 0 and not g['grad'][4].any()
    assert np.array_equal(np.isfinite(loss), np.isfinite(g['loss']))
    fin = np.isfinite(loss)
    assert close(loss[fin], g['loss'][fin], 1e-12) and close(grad, g['grad'])
    dense, dl = O.ctc_greedy_decode(g['logits'], g['len_logits'])
    assert np.array_equal(dense, g['greedy']) and np.array_equal(dl, g['greedy_len'])


def test_shrink():
    g = load('shrink')
    out, new_len, flag = O.shrink_fwd(g['x'], g['lens'], g['logits'])
    assert np.array_equal(out, g['out']) and np.array_equal(new_len, g['new_len']) and new_len[2] == 0
    assert np.array_equal(O.shrink_bwd(g['dout'], flag, g['x'].shape[1]), g['dx'])


def test_xent_mse_adam_lr():
    g = load('xent_mse_adam')
    loss, grad = O.ls_xent_fwd_bwd(g['logits'].astype(np.float64), g['labels'], g['lens'], 0.9)
    assert close(loss, g['xent'], 1e-12) and close(grad, g['dlogits'])
    mse, da = O.mse_fwd_bwd(g['b'].astype(np.float64), g['a'].astype(np.float64))
    assert close(mse, g['mse'], 1e-12) and close(da, g['da'])
    p2, m2, v2 = O.adam_step(g['p'].astype(np.float64), g['g'].astype(np.float64), g['m'].astype(np.float64), g['v'].astype(np.float64),
                             float(g['lr']), int(g['t']))
    assert close(p2, g['p2']) and close(m2, g['m2']) and close(v2, g['v2'])
    for s, want in zip(g['lr_steps'], g['lr_values']):
        assert O.warmup_exponential_decay(int(s), 10, 1e-3, 0.5, 100) == want


@pytest.mark.parametrize('name,loss_type,shrink', [('model_lut', 'ctc_kd_ce', False), ('model_costt', 'ce_ctc', True)])
def test_model_step_and_greedy(name, loss_type, shrink):
    g = load(name)
    cfg = tiny_cfg(loss_type, shrink).zero_dropout()
    P = {k[6:]: v.astype(np.float64) for k, v in g.items() if k.startswith('param/')}
    batch = {k[6:]: v for k, v in g.items() if k.startswith('batch/')}
    feat, lens = O.splice_downsample_batch(batch['feature'], batch['len_fea'], 5, 3)
    ref = OracleTPE(cfg, is_train=True, dropout=False).loss_and_grads(P, dict(batch, feature=feat, len_fea=lens))
    assert close(ref['loss'], g['loss'], 1e-12) and close(ref['loss_ctc'], g['loss_ctc'], 1e-12) and close(ref['loss_ast'], g['loss_ast'], 1e-12)
    assert close(ref['logits'], g['logits']) and close(ref['logits_am'], g['logits_am'])
    for k, v in ref['grads'].items():
        assert close(v, g['grad/' + k]), k
    ids, ids_am, dec_len = OracleTPE(cfg, is_train=False).infer(P, feat, lens)
    assert np.array_equal(ids, g['greedy']) and np.array_equal(ids_am, g['greedy_am']) and np.array_equal(dec_len, g['greedy_len'])
