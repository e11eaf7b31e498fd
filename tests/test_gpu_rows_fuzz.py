"""-m gpu: seeded random walks over the row kernels (residual + LayerNorm, CTC loss / gradient / greedy decode, label-smoothed
cross entropy, shrink) through the C ABI, against the numpy oracle (oracle/ops.py; reference: layer_norm /
layer_postprocess st/layers/common_layers_v1.py:342-380, tf.nn.ctc_loss as called in st/models/lstmModel.py, shrink_layer
st/models/tools/blocks.py:8-38, smoothing_cross_entropy st/layers/common_layers_v1.py).

The fixed cases of test_gpu_ops.py pin the degenerate inputs; here widths, row counts, lengths (zero included), dropout rates
and dtypes are drawn independently so that every dispatch path (straight-line / 16-byte / 8-byte / scalar LayerNorm lanes,
register-resident / streaming CTC rows, one- and multi-wave label counts) sees shapes nobody wrote down."""
import os

import numpy as np
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu
FUZZ_SCALE = int(os.environ.get('NEURST_FUZZ_SCALE', '1'))     # more seeds for a longer hunt (not the committed suite)

if torch.cuda.is_available():
    from neurst_amd import ops
    DEV = torch.device('cuda:0')


def _bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).float().numpy()


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.to(dtype) if dtype is not None else t


def _host(t):
    return t.float().cpu().numpy() if t.is_floating_point() else t.cpu().numpy()


def _relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


@pytest.mark.parametrize('seed', range(36 * FUZZ_SCALE))
def test_residual_layernorm_random_shapes(seed):
    rng = np.random.default_rng(5100 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]
    dtype = 'f32' if seed % 3 == 0 else 'bf16'
    C = pick([768, 512, 256, 1024, 8, 24, 100, 40, 192, 384, 1536])
    B, T = int(rng.integers(1, 6)), pick([1, 2, 17, 64, 65, 150, 333])
    rate = pick([0.0, 0.1, 0.3])
    masked = bool(rng.integers(0, 2))            # without a row mask bf16 rows of 256 / 512 / 768 take the straight-line kernels
    deterministic = bool(rng.integers(0, 2))
    td, tol = (torch.float32, 1e-5) if dtype == 'f32' else (torch.bfloat16, 2e-2)
    x, y, dout = (rng.standard_normal((B, T, C)).astype(np.float32) for _ in range(3))
    if dtype == 'bf16':
        x, y, dout = _bf16(x), _bf16(y), _bf16(dout)
    gam = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    bet = (0.1 * rng.standard_normal(C)).astype(np.float32)
    lens = rng.integers(0, T + 1, size=B).astype(np.int32)
    lens[0] = T
    keep = (np.arange(T)[None, :] < lens[:, None]).astype(np.float32)[..., None] if masked else np.ones((B, T, 1), np.float32)
    key = O.site_key(7, 100 + seed, 0)
    what = dict(seed=seed, dtype=dtype, B=B, T=T, C=C, rate=rate, masked=masked, deterministic=deterministic, lens=lens.tolist())
    ref, cache = O.residual_fwd(x, y, gam, bet, rate, key)
    dz_ref, dy_ref, dg_ref, db_ref = O.residual_bwd(dout * keep, cache, gam)
    yz = _dev(y, td)
    args = (_dev(lens), T) if masked else ()
    out, mean, rstd = ops.dropout_add_layernorm_fwd(_dev(x, td), yz, _dev(gam), _dev(bet), rate, key, *args)
    assert _relerr(_host(out), ref * keep) < tol, ('out', what)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dz, dy = ops.dropout_add_layernorm_bwd(_dev(dout, td), yz, _dev(gam), mean, rstd, dg, db, rate, key, *args, deterministic=deterministic)
    for name, got, want in (('dz', dz, dz_ref), ('dy', dy, dy_ref), ('dgamma', dg, dg_ref), ('dbeta', db, db_ref)):
        assert _relerr(_host(got), want) < tol, (name, _relerr(_host(got), want), what)


@pytest.mark.parametrize('seed', range(24 * FUZZ_SCALE))
def test_ctc_random_batches(seed):
    """loss, gradient (fp32, scaled, in place, bf16) and the greedy path of random batches: repeated labels, utterances shorter
    than their label sequence (ignored: loss 0, gradient 0), label counts on both sides of one wavefront's 2U + 1 <= 64 states"""
    rng = np.random.default_rng(7300 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]
    V = pick([5, 37, 40, 100, 1000, 1001, 2048, 10000])
    B, T = int(rng.integers(1, 7)), pick([1, 2, 9, 50, 130, 257])
    U = pick([1, 3, 7, 31, 32, 40, 70])
    logits = (rng.standard_normal((B, T, V)) * 2).astype(np.float32)
    hi = max(V - 1, 1)
    labels = rng.integers(0, hi, size=(B, U)).astype(np.int32)
    rep = rng.random((B, U)) < 0.3                          # runs of repeated labels need a blank between them
    for u in range(1, U):
        labels[:, u] = np.where(rep[:, u], labels[:, u - 1], labels[:, u])
    len_labels = rng.integers(1, U + 1, size=B).astype(np.int32)      # (empty sequences: tf.nn.ctc_loss rejects them; not drawn)
    len_logits = rng.integers(1, T + 1, size=B).astype(np.int32)
    len_logits[0], len_labels[0] = T, min(U, max(T // 2, 1) if T > 1 else 1)
    scale = pick([1.0, 0.25])
    what = dict(seed=seed, B=B, T=T, V=V, U=U, len_logits=len_logits.tolist(), len_labels=len_labels.tolist(), scale=scale)
    ref_loss, ref_grad = O.ctc_loss_fwd_bwd(logits.astype(np.float64), len_logits, labels, len_labels)
    loss, grad, am = ops.ctc_loss_fwd_bwd(_dev(logits), _dev(len_logits), _dev(labels), _dev(len_labels), ops.F32, scale)
    loss, grad, am = _host(loss), _host(grad), _host(am)
    fin = np.isfinite(ref_loss)
    assert np.array_equal(np.isposinf(loss), np.isposinf(ref_loss)), what
    if fin.any():
        assert _relerr(loss[fin], ref_loss[fin]) < 1e-5, what
    assert _relerr(grad, scale * ref_grad) < 1e-5 or np.abs(ref_grad).max() == 0 and not grad.any(), what
    live = np.arange(T)[None, :] < len_logits[:, None]
    assert np.array_equal(am[live], logits.argmax(-1)[live]), what
    assert not grad[~live].any(), ('padded frames carry gradient', what)
    dense, dl = O.ctc_greedy_decode(logits, len_logits)
    got_d, got_l = ops.ctc_greedy(_dev(am.astype(np.int32)), _dev(len_logits), V - 1)
    assert np.array_equal(_host(got_l), dl) and np.array_equal(_host(got_d)[:, :dense.shape[1]], dense), what
    lg = _dev(logits)
    _, g2, _ = ops.ctc_loss_fwd_bwd(lg, _dev(len_logits), _dev(labels), _dev(len_labels), ops.F32, scale, in_place=True)
    assert g2.data_ptr() == lg.data_ptr() and np.abs(_host(g2) - grad).max() <= 1e-5 * max(np.abs(grad).max(), 1e-30), what
    _, g3, _ = ops.ctc_loss_fwd_bwd(_dev(logits), _dev(len_logits), _dev(labels), _dev(len_labels), ops.BF16, scale)
    assert np.abs(_host(g3) - scale * ref_grad).max() <= 1e-2 * max(np.abs(ref_grad).max() * scale, 1e-12) + 1e-30, what


@pytest.mark.parametrize('seed', range(16 * FUZZ_SCALE))
def test_ls_xent_random_batches(seed):
    rng = np.random.default_rng(8800 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]
    V = pick([2, 11, 37, 1000, 1001, 4096, 10000])
    B, L = int(rng.integers(1, 7)), pick([1, 7, 40, 81])
    conf = pick([0.9, 1.0, 0.7])
    scale = pick([1.0, 0.5])
    logits = (rng.standard_normal((B, L, V)) * 3).astype(np.float32)
    labels = rng.integers(0, V, size=(B, L)).astype(np.int32)
    lens = rng.integers(1, L + 1, size=B).astype(np.int32)     # (a target holds at least its end-of-sentence label; the
    lens[0] = L                                                # reference's per-utterance mean over zero tokens is 0 / 0)
    what = dict(seed=seed, B=B, L=L, V=V, conf=conf, lens=lens.tolist())
    ref_loss, ref_grad = O.ls_xent_fwd_bwd(logits.astype(np.float64), labels, lens, conf)
    xent, grad = ops.ls_xent_fwd_bwd(_dev(logits), _dev(labels), _dev(lens), conf, ops.F32, scale)
    assert _relerr(_host(xent).sum(-1), ref_loss) < 1e-5, what
    assert _relerr(_host(grad), scale * ref_grad) < 1e-5, what


@pytest.mark.parametrize('seed', range(16 * FUZZ_SCALE))
def test_shrink_random_batches(seed):
    """frames whose greedy CTC label is blank or repeats its predecessor are dropped (blocks.py:8-38): index, new lengths,
    gathered rows and their scatter back are bit-exact, for blank-heavy and label-heavy logits, zero-length utterances included"""
    rng = np.random.default_rng(9900 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]
    B, T, D, V = int(rng.integers(1, 9)), pick([1, 5, 64, 300, 513]), pick([8, 24, 512]), pick([3, 9, 100])
    logits = rng.standard_normal((B, T, V)).astype(np.float32)
    logits[..., V - 1] += pick([-2.0, 0.0, 1.0, 3.0])      # how often the blank wins
    if B > 1:
        logits[B - 1, :, V - 1] += 100.0                    # an utterance with no kept frame
    x = rng.standard_normal((B, T, D)).astype(np.float32)
    lens = rng.integers(0, T + 1, size=B).astype(np.int32)
    lens[0] = T
    what = dict(seed=seed, B=B, T=T, D=D, V=V, lens=lens.tolist())
    ref, ref_len, flag = O.shrink_fwd(x, lens, logits)
    am = _dev(logits.argmax(-1).astype(np.int32))
    index, new_len = ops.shrink_index(am, _dev(lens), V - 1)
    assert np.array_equal(_host(new_len), ref_len), what
    out = ops.shrink_fwd(_dev(x), index, ref.shape[1])
    assert np.array_equal(_host(out), ref), what
    dout = rng.standard_normal(ref.shape).astype(np.float32)
    dout *= (np.arange(ref.shape[1])[None, :, None] < ref_len[:, None, None])
    dx = ops.shrink_bwd(_dev(dout), index, T)
    assert np.array_equal(_host(dx), O.shrink_bwd(dout, flag, T)), what
