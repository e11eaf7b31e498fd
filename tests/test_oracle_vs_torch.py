"""Cross-check of the numpy oracle (hand-derived backward) against an independent
torch-CPU autograd expression of the same graph."""
import numpy as np
import pytest
import torch

from oracle import ops
from oracle.model import OracleTPE, TPEConfig, init_params
from neurst_amd.synthetic import make_batch
from oracle import torch_graph as torch_ref


def tiny_cfg(**kw):
    c = TPEConfig(dim_input=24, dim_output=11, enc_blocks=2, enc_heads=2, enc_units=16, enc2_blocks=2, enc2_heads=2,
                  enc2_units=16, dec_blocks=2, dec_heads=2, dec_units=16, size_embedding=16, kd_units=16, kd_heads=2,
                  loss_type='ctc_kd_ce').zero_dropout()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def run_both(cfg, batch, seed=3):
    P = init_params(cfg, seed=seed, dtype=np.float64, perturb=True)
    out = OracleTPE(cfg, is_train=True, dropout=False).loss_and_grads(P, batch)
    Pt = {k: torch.tensor(v, requires_grad=True) for k, v in P.items()}
    loss, aux = torch_ref.loss_fn(Pt, cfg, batch)
    loss.backward()
    return P, out, Pt, loss, aux


def check(P, out, Pt, loss, aux, tol=1e-9):
    assert abs(out['loss'] - loss.item()) <= tol * max(1, abs(loss.item()))
    np.testing.assert_allclose(out['logits'], aux['logits'].detach().numpy(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(out['logits_am'], aux['logits_am'].detach().numpy(), rtol=1e-9, atol=1e-10)
    for k in P:
        g = Pt[k].grad
        g = np.zeros_like(P[k]) if g is None else g.numpy()
        scale = max(np.abs(g).max(), 1e-12)
        assert np.abs(out['grads'][k] - g).max() <= 1e-8 * scale + 1e-13, k


def test_lut_tiny_matches_torch():
    cfg = tiny_cfg()
    batch = make_batch(B=3, Traw=12, V=11, L=5, U=4, Lb=6, bert_dim=16, raw=False, dim_input=24, ragged=True, dtype=np.float64)
    check(*run_both(cfg, batch))


def test_costt_tiny_matches_torch():
    cfg = tiny_cfg(loss_type='ce_ctc', shrink_layer=True)
    batch = make_batch(B=3, Traw=14, V=11, L=7, U=4, raw=False, dim_input=24, ragged=True, dtype=np.float64, seed=7)
    P, out, Pt, loss, aux = run_both(cfg, batch)
    assert out['len_enc'].max() >= 1
    check(P, out, Pt, loss, aux)


def test_cnn_frontend_matches_torch():
    cfg = tiny_cfg(use_cnn_resample=True, cnn_layers=2, num_filters=4, cnn_kernel=(3, 3), cnn_stride=(2, 2),
                   feature_channel=1, loss_type='ce_ctc')
    batch = make_batch(B=2, Traw=21, V=11, L=5, U=3, raw=False, dim_input=24, ragged=True, dtype=np.float64, seed=5)
    check(*run_both(cfg, batch))


def test_ctc_matches_torch_with_repeats_and_infeasible():
    rng = np.random.default_rng(0)
    B, T, V = 5, 9, 6
    logits = rng.standard_normal((B, T, V))
    labels = np.array([[1, 1, 2, 0], [3, 2, 3, 2], [4, 0, 0, 0], [1, 2, 3, 4], [2, 2, 2, 2]], np.int32)
    len_labels = np.array([3, 4, 1, 4, 4], np.int32)
    len_logits = np.array([9, 7, 5, 3, 9], np.int32)      # row 3: U > T  -> loss 0, grad 0
    loss, grad = ops.ctc_loss_fwd_bwd(logits, len_logits, labels, len_labels)
    z = torch.tensor(logits, requires_grad=True)
    lt = torch.nn.functional.ctc_loss(torch.log_softmax(z, -1).transpose(0, 1), torch.tensor(labels).long(),
                                      torch.tensor(len_logits).long(), torch.tensor(len_labels).long(), blank=V - 1,
                                      reduction='none', zero_infinity=True)
    lt.sum().backward()
    np.testing.assert_allclose(loss, lt.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(grad, z.grad.numpy(), rtol=1e-8, atol=1e-10)
    assert loss[3] == 0 and np.all(grad[3] == 0)
    assert np.all(grad[1, 7:] == 0)


def test_ctc_no_valid_path_is_inf_with_softmax_grad():
    # U <= T but U + repeats > T: TF yields loss = +inf ("No valid path found") and dy = y
    rng = np.random.default_rng(1)
    logits = rng.standard_normal((1, 3, 5))
    loss, grad = ops.ctc_loss_fwd_bwd(logits, np.array([3]), np.array([[1, 1, 1]]), np.array([3]))
    assert np.isposinf(loss[0])
    e = np.exp(logits[0] - logits[0].max(-1, keepdims=True))
    np.testing.assert_allclose(grad[0], e / e.sum(-1, keepdims=True), rtol=1e-12)


def test_finite_difference_end_to_end():
    cfg = tiny_cfg()
    batch = make_batch(B=2, Traw=8, V=11, L=4, U=3, Lb=3, bert_dim=16, raw=False, dim_input=24, dtype=np.float64)
    P = init_params(cfg, seed=2, dtype=np.float64, perturb=True)
    model = OracleTPE(cfg, is_train=True, dropout=False)
    g = model.loss_and_grads(P, batch)['grads']
    rng = np.random.default_rng(0)
    for name in ['transformer/encoder/encoder_fc/kernel', 'transformer/embedding',
                 'transformer/decoder_vanilla_attention/kv_transform/kernel',
                 'transformer/TPEncoder/block_1/layer_norm/layer_norm_scale',
                 'transformer/decoder/decoder/fully_connected/kernel']:
        d = rng.standard_normal(P[name].shape)
        eps = 1e-6
        Pp = dict(P); Pp[name] = P[name] + eps * d
        Pm = dict(P); Pm[name] = P[name] - eps * d
        fd = (model.loss_and_grads(Pp, batch)['loss'] - model.loss_and_grads(Pm, batch)['loss']) / (2 * eps)
        an = (g[name] * d).sum()
        assert abs(fd - an) <= 1e-6 * max(1.0, abs(an)), (name, fd, an)


def test_dropout_masks_are_reproducible_and_unbiased():
    key = ops.site_key(5, 102, 7)
    m1 = ops.keep_mask(key, 200000, 0.1)
    m2 = ops.keep_mask(key, 200000, 0.1)
    assert np.array_equal(m1, m2)
    assert abs(m1.mean() - 0.9) < 5e-3
    assert ops.keep_mask(ops.site_key(5, 103, 7), 1000, 0.1).tolist() != m1[:1000].tolist()


def test_attention_keep_words_follow_their_stated_rule():
    """attn_keep_words (32 decisions per word, bit-sliced 16-bit comparison) against a scalar restatement of the rule in its
    comment, and its statistics: P(keep) = t16 / 65536, decisions inside a word uncorrelated, drops per word binomial."""
    key, rate, nw = ops.site_key(3, 41, 9), 0.1, 3
    words = ops.attn_keep_words(key, 5, nw, rate)
    t16 = ops.attn_threshold16(rate)
    assert t16 == 58982 and ops.attn_threshold16(0.5) == 32768

    def fm(x):
        x &= 0xFFFFFFFF; x ^= x >> 16; x = (x * 0x85EBCA6B) & 0xFFFFFFFF; x ^= x >> 13; x = (x * 0xC2B2AE35) & 0xFFFFFFFF; x ^= x >> 16
        return x

    def xs(x):
        x ^= (x << 13) & 0xFFFFFFFF; x ^= x >> 17; x ^= (x << 5) & 0xFFFFFFFF
        return x
    for widx in (0, 7, 14):
        W = []
        for k in range(16):
            v = fm((4 * widx + (k & 3)) ^ fm(0 ^ key))
            for _ in range(k >> 2):
                v = xs(v)
            W.append(v)
        for i in range(32):
            U = sum((1 - ((W[k] >> i) & 1)) << k for k in range(16))
            assert bool((int(words.reshape(-1)[widx]) >> i) & 1) == (U < t16)
    m = ops.attn_keep_mask(key, 100000, 64, rate).astype(np.float64)
    assert m.shape == (100000, 64) and abs(m.mean() - t16 / 65536.0) < 1e-3
    c = np.corrcoef(m.T)
    np.fill_diagonal(c, 0)
    assert np.abs(c).max() < 0.02
    drops = (1 - m[:, :32]).sum(1)
    assert abs(drops.var() - 32 * 0.1 * 0.9) < 0.1
    assert not np.array_equal(ops.attn_keep_words(key + 1, 5, nw, rate), words)


def test_greedy_infer_runs_and_matches_teacher_forcing_prefix():
    cfg = tiny_cfg(max_len=12)
    P = init_params(cfg, seed=4, dtype=np.float64, perturb=True)
    batch = make_batch(B=2, Traw=9, V=11, L=5, U=3, Lb=3, bert_dim=16, raw=False, dim_input=24, dtype=np.float64)
    m = OracleTPE(cfg, is_train=False)
    preds, preds_am, lens = m.infer(P, batch['feature'], batch['len_fea'])
    assert preds.shape[0] == 2 and preds.shape[1] == lens.max()
    # teacher-forcing the greedy output must reproduce it (causal decoder == cached decoder)
    lab = preds.copy()
    b2 = dict(batch, label=lab, len_label=np.maximum(lens, 1))
    out = OracleTPE(cfg, is_train=True, dropout=False).loss_and_grads(P, b2)
    tf_preds = out['logits'].argmax(-1)
    for b in range(2):
        assert np.array_equal(tf_preds[b, :lens[b]], preds[b, :lens[b]])
