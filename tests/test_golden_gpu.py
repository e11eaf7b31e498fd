"""-m gpu: the HIP path (through the C ABI / host mirror) against the committed golden vectors only - this
file imports nothing from oracle/.  fp32 mode, 1e-5 relative (2e-5 where a softmax chain is involved);
integer outputs bit-exact."""
import os

import numpy as np
import pytest
import torch

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

if torch.cuda.is_available():
    from neurst_amd import ops
    DEV = torch.device('cuda:0')


def load(name):
    return dict(np.load(os.path.join(G, name + '.npz')))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.float().cpu().numpy() if t.is_floating_point() else t.cpu().numpy()


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def test_splice():
    g = load('splice')
    out, olen = ops.splice_downsample(dev(g['x']), dev(g['lens']), 5, 3, ops.F32)
    assert np.array_equal(host(out), g['out']) and np.array_equal(host(olen), g['out_len'])
    # the reference's docstring example: right context 2, no down-sampling
    out, _ = ops.splice_downsample(dev(g['doc_in'][None]), dev(np.array([7], np.int32)), 2, 1, ops.F32)
    assert np.array_equal(host(out)[0], g['doc_want'])


@pytest.mark.parametrize('tag', ['r0', 'r25'])
def test_layernorm(tag):
    g = load('layernorm')
    rate, key = float(g[tag + '_rate']), int(g[tag + '_key'])
    y = dev(g['y'])
    out, mean, rstd = ops.dropout_add_layernorm_fwd(dev(g['x']), y, dev(g['gamma']), dev(g['beta']), rate, key)
    assert relerr(host(out), g[tag + '_out']) < 1e-5
    C = g['x'].shape[-1]
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dz, dy = ops.dropout_add_layernorm_bwd(dev(g['dout']), y, dev(g['gamma']), mean, rstd, dg, db, rate, key)
    for got, k in ((dz, 'dz'), (dy, 'dy'), (dg, 'dgamma'), (db, 'dbeta')):
        assert relerr(host(got), g[tag + '_' + k]) < 1e-5, k


@pytest.mark.parametrize('name', ['attention_tiny', 'attention_causal', 'attention_dh96'])
def test_attention(name):
    g = load(name)
    B, Tq, d = g['q'].shape
    Tk, H, mode = g['k'].shape[1], int(g['H']), int(g['mode'])
    dh = d // H
    q, k, v = dev(g['q']), dev(g['k']), dev(g['v'])
    lens = dev(g['lens']) if mode == ops.MASK_KEYPAD else None
    o, lse = ops.mha_fwd(q, k, v, d, d, d, B, H, Tq, Tk, dh, mode, lens, float(g['rate']), int(g['key']))
    assert relerr(host(o), g['o']) < 2e-5
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.mha_bwd(q, k, v, o, dev(g['do']), lse, dq, dk, dv, d, d, d, d, d, d, B, H, Tq, Tk, dh, mode, lens, float(g['rate']), int(g['key']))
    assert relerr(host(dq), g['dq']) < 2e-5 and relerr(host(dk), g['dk']) < 2e-5 and relerr(host(dv), g['dv']) < 2e-5


def test_ctc_edge_rows_and_greedy():
    g = load('ctc')
    V = g['logits'].shape[-1]
    loss, grad, am = ops.ctc_loss_fwd_bwd(dev(g['logits']), dev(g['len_logits']), dev(g['labels']), dev(g['len_labels']), ops.F32, 1.0)
    loss, grad = host(loss), host(grad)
    assert np.isposinf(loss[3]) and loss[4] == 0 and not grad[4].any()
    fin = np.isfinite(g['loss'])
    assert relerr(loss[fin], g['loss'][fin]) < 1e-5 and relerr(grad, g['grad']) < 1e-5
    out, out_len = ops.ctc_greedy(am.to(torch.int32), dev(g['len_logits']), V - 1)
    n = g['greedy'].shape[1]
    assert np.array_equal(host(out_len), g['greedy_len']) and np.array_equal(host(out)[:, :n], g['greedy']) and not host(out)[:, n:].any()


def test_shrink():
    g = load('shrink')
    V = g['logits'].shape[-1]
    B, T, _ = g['logits'].shape
    am = ops.argmax_rows(dev(g['logits']).view(B * T, V)).view(B, T)
    index, new_len = ops.shrink_index(am, dev(g['lens']), V - 1)
    assert np.array_equal(host(new_len), g['new_len'])
    assert np.array_equal(host(ops.shrink_fwd(dev(g['x']), index, g['out'].shape[1])), g['out'])
    assert np.array_equal(host(ops.shrink_bwd(dev(g['dout']), index, g['x'].shape[1])), g['dx'])


def test_xent_mse_adam():
    g = load('xent_mse_adam')
    xent, grad = ops.ls_xent_fwd_bwd(dev(g['logits']), dev(g['labels']), dev(g['lens']), 0.9, ops.F32, 1.0)
    assert relerr(host(xent).sum(-1), g['xent']) < 1e-5 and relerr(host(grad), g['dlogits']) < 1e-5
    partial, da = ops.mse_fwd_bwd(dev(g['a']), dev(g['b']), 1.0)
    assert relerr(host(da), g['da']) < 1e-5
    out = host(ops.loss_combine(None, None, partial, g['a'].size, 1, 1, 0.0, 0.0, 1.0, DEV))
    assert abs(out[3] - g['mse']) < 1e-5 * g['mse']
    p, m, v = dev(g['p']), dev(g['m']), dev(g['v'])
    import math
    t = int(g['t'])
    lr_t = float(g['lr']) * math.sqrt(1 - 0.98 ** t) / (1 - 0.9 ** t)        # TF folds the bias correction into the step size
    ops.adam_step(p, dev(g['g']), m, v, None, lr_t, 0.9, 0.98, 1e-9, 1.0)
    assert relerr(host(p), g['p2']) < 1e-6 and relerr(host(m), g['m2']) < 1e-6 and relerr(host(v), g['v2']) < 1e-6


@pytest.mark.parametrize('name,over', [('model_lut', dict(loss_type='ctc_kd_ce')), ('model_costt', dict(loss_type='ce_ctc', shrink=True))])
def test_model_step_and_greedy(name, over):
    from neurst_amd.st.models.TransformerTPE import GlobalStep
    g = load(name)
    args = ge.tiny_args(compute_dtype='fp32', **over)
    P = {k[6:]: v for k, v in g.items() if k.startswith('param/')}
    batch = {k[6:]: v for k, v in g.items() if k.startswith('batch/')}
    model = args.Model(GlobalStep(0), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
    model.keep_logits = True
    model.store.load_params(P)
    out4 = model.forward_backward(batch).cpu().numpy()
    assert abs(out4[0] - g['loss']) < 1e-5 * abs(g['loss'])
    assert abs(out4[1] - g['loss_ast']) < 1e-5 * abs(g['loss_ast']) and abs(out4[2] - g['loss_ctc']) < 1e-5 * abs(g['loss_ctc'])
    assert relerr(model.last['logits'].cpu().numpy(), g['logits']) < 1e-5
    assert relerr(model.last['logits_am'].cpu().numpy(), g['logits_am']) < 1e-5
    assert np.array_equal(model.last['len_encoded'].cpu().numpy(), g['len_enc'])
    grads = model.store.export_grads()
    for k in P:
        assert relerr(grads[k], g['grad/' + k]) < 5e-5, k
    infer = args.Model(GlobalStep(0), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=False, args=args)
    infer.store.load_params(P)
    preds, preds_am = infer.infer(batch['feature'], batch['len_fea'])
    assert np.array_equal(preds, g['greedy']) and np.array_equal(preds_am, g['greedy_am'])
    assert np.array_equal(infer.last['len_decoded'], g['greedy_len'])
