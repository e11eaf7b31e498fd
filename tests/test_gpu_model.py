"""-m gpu: the whole TransformerTPE train step / greedy decode through the host mirror + C ABI
against the numpy oracle with identical weights."""
import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from oracle import ops as O
from oracle.model import OracleTPE, train_step as oracle_train_step
from neurst_amd.synthetic import make_batch

pytestmark = pytest.mark.gpu


def make_model(args, is_train=True, step=0):
    from neurst_amd.st.models.TransformerTPE import GlobalStep
    gs = GlobalStep(step)
    return args.Model(gs, encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=is_train, args=args)


def perturb(model, seed=0):
    """randomise biases / LN parameters too, so their gradients are exercised"""
    rng = np.random.default_rng(seed)
    P = model.store.export()
    for k, v in P.items():
        if v.ndim == 1:
            P[k] = (v + 0.1 * rng.standard_normal(v.shape)).astype(np.float32)
    model.store.load_params(P)
    return P


def oracle_batch(batch, nctx=5, rate=3):
    feat, lens = O.splice_downsample_batch(batch['feature'], batch['len_fea'], nctx, rate)
    return dict(batch, feature=feat, len_fea=lens)


def compare_grads(model, ref, tol):
    g = model.store.export_grads()
    worst = 0.0
    for k, v in ref['grads'].items():
        err = np.abs(g[k] - v).max() / max(np.abs(v).max(), 1e-7)
        worst = max(worst, err)
        assert err < tol, (k, err)
    return worst


CASES = {
    'lut': dict(loss_type='ctc_kd_ce'),
    'costt': dict(loss_type='ce_ctc', shrink=True),
}


@pytest.mark.parametrize('case', ['lut', 'costt'])
def test_fp32_losses_logits_grads_match_oracle(case):
    args = ge.tiny_args(compute_dtype='fp32', **CASES[case])
    model = make_model(args)
    model.keep_logits = True
    P = perturb(model)
    batch = make_batch(B=3, Traw=40, V=11, L=6, U=3, Lb=5, bert_dim=16, dim_raw=4, ragged=True, seed=case == 'lut')
    out4 = model.forward_backward(batch).cpu().numpy()
    cfg = ge.oracle_cfg_from_args(args).zero_dropout()
    ref = OracleTPE(cfg, is_train=True, dropout=False).loss_and_grads({k: v.astype(np.float64) for k, v in P.items()}, oracle_batch(batch))
    assert abs(out4[0] - ref['loss']) < 1e-5 * abs(ref['loss'])
    assert abs(out4[1] - ref['loss_ast']) < 1e-5 * abs(ref['loss_ast'])
    assert abs(out4[2] - ref['loss_ctc']) < 1e-5 * abs(ref['loss_ctc'])
    if case == 'lut':
        assert abs(out4[3] - ref['loss_bert']) < 1e-5 * abs(ref['loss_bert'])
    lg = model.last['logits'].cpu().numpy()
    assert np.abs(lg - ref['logits']).max() < 1e-5 * np.abs(ref['logits']).max()
    lam = model.last['logits_am'].cpu().numpy()
    assert np.abs(lam - ref['logits_am']).max() < 1e-5 * np.abs(ref['logits_am']).max()
    compare_grads(model, ref, 5e-5)
    assert np.array_equal(model.last['len_encoded'].cpu().numpy(), ref['len_enc'])


@pytest.mark.parametrize('case,B,Traw,L,U,Lb', [('lut', 1, 3, 1, 1, 1), ('lut', 2, 1, 2, 1, 2), ('costt', 1, 2, 1, 1, 1), ('costt', 2, 7, 3, 5, 1)])
def test_fp32_degenerate_shapes_match_oracle(case, B, Traw, L, U, Lb):
    """smallest shapes: one utterance, one encoder frame, one label; CTC rows whose label is longer than the input
    (U = 5 on T = 3: infeasible, zeroed like `ignore_longer_outputs_than_inputs`), shrink with nothing kept"""
    args = ge.tiny_args(compute_dtype='fp32', **CASES[case])
    model = make_model(args)
    P = perturb(model, seed=B + Traw)
    batch = make_batch(B=B, Traw=Traw, V=11, L=L, U=U, Lb=Lb, bert_dim=16, dim_raw=4, ragged=False, seed=3)
    out4 = model.forward_backward(batch).cpu().numpy()
    cfg = ge.oracle_cfg_from_args(args).zero_dropout()
    ref = OracleTPE(cfg, is_train=True, dropout=False).loss_and_grads({k: v.astype(np.float64) for k, v in P.items()}, oracle_batch(batch))
    assert np.isfinite(out4).all()
    assert abs(out4[0] - ref['loss']) < 1e-5 * max(abs(ref['loss']), 1e-3), (out4, ref['loss'])
    assert abs(out4[2] - ref['loss_ctc']) < 1e-5 * max(abs(ref['loss_ctc']), 1e-3)
    # with one key the attention weights are exactly 1, so the query / key projections have an exactly zero gradient:
    # those tensors are compared on an absolute floor (1e-3, far below a typical gradient entry) instead of their own maximum
    g = model.store.export_grads()
    for k, v in ref['grads'].items():
        err = np.abs(g[k] - v).max() / max(np.abs(v).max(), 1e-3)
        assert err < 5e-5, (k, err)
    assert np.array_equal(model.last['len_encoded'].cpu().numpy(), ref['len_enc'])


def test_fp32_cnn_frontend_matches_oracle():
    args = ge.tiny_args(compute_dtype='fp32', loss_type='ce_ctc', use_cnn_resample=True, cnn={'layers': 2, 'kernel': '3,3', 'stride': '2,2'},
                        num_filters=4)
    model = make_model(args)
    P = perturb(model)
    batch = make_batch(B=2, Traw=21, V=11, L=5, U=3, raw=False, dim_input=24, ragged=True, seed=5)
    out4 = model.forward_backward(batch).cpu().numpy()
    cfg = ge.oracle_cfg_from_args(args).zero_dropout()
    ref = OracleTPE(cfg, is_train=True, dropout=False).loss_and_grads({k: v.astype(np.float64) for k, v in P.items()}, batch)
    assert abs(out4[0] - ref['loss']) < 1e-5 * abs(ref['loss'])
    compare_grads(model, ref, 1e-4)


def test_bf16_cnn_frontend_fast_paths_within_tolerance():
    """use_cnn_resample in bf16 with 32 filters: the first layer takes the one-K-step MFMA kernel (kh kw Cin = 9), the second the
    3x3 implicit-GEMM kernels (forward, data gradient, im2col weight gradient), the max pool its vector path.  The step runs
    once with the fast paths and once with the general kernels (neurst_set_option conv_fast 0): loss against the fp64 oracle
    at 2e-2, every gradient at 0.3 of its largest element (d_model 16 in bf16: the general kernels sit at 0.27 on the decoder's
    query bias), and the two kernel sets within 0.03 of each other, which is what pins the fast paths inside the model."""
    from neurst_amd import ops
    args = ge.tiny_args(compute_dtype='bf16', loss_type='ce_ctc', use_cnn_resample=True, cnn={'layers': 2, 'kernel': '3,3', 'stride': '2,2'},
                        num_filters=32)
    batch = make_batch(B=2, Traw=37, V=11, L=5, U=3, raw=False, dim_input=24, ragged=True, seed=5)
    got = {}
    try:
        for fast in (1, 0):
            ops.set_option('conv_fast', fast)
            model = make_model(args)
            if fast:
                P = perturb(model)
                P = {k: (torch.from_numpy(v).bfloat16().float().numpy() if v.ndim > 1 and 'embedding' not in k else v) for k, v in P.items()}
                cfg = ge.oracle_cfg_from_args(args).zero_dropout()
                ref = OracleTPE(cfg, is_train=True, dropout=False).loss_and_grads({k: v.astype(np.float64) for k, v in P.items()}, batch)
            model.store.load_params(P)
            out4 = model.forward_backward(batch).cpu().numpy()
            assert abs(out4[0] - ref['loss']) < 2e-2 * abs(ref['loss'])
            compare_grads(model, ref, 0.3)
            got[fast] = (out4, model.store.export_grads())
    finally:
        ops.set_option('conv_fast', 1)
    assert abs(got[1][0][0] - got[0][0][0]) < 2e-3 * abs(ref['loss'])
    for k, v in got[0][1].items():
        err = np.abs(got[1][1][k] - v).max() / max(np.abs(v).max(), 1e-7)
        assert err < 0.03, (k, err)


def test_fp32_dropout_and_specaug_parity_with_oracle_hash():
    over = dict(loss_type='ce_ctc', use_specAug=True)
    args = ge.tiny_args(compute_dtype='fp32', **over)
    for part in ('encoder', 'encoder2', 'decoder'):
        getattr(args.model, part).attention_dropout_rate = 0.1
        getattr(args.model, part).residual_dropout_rate = 0.1
    args.seed = 17
    model = make_model(args, step=5)
    P = perturb(model)
    batch = make_batch(B=3, Traw=150, V=11, L=6, U=3, dim_raw=4, ragged=True, seed=3)
    out4 = model.forward_backward(batch).cpu().numpy()
    cfg = ge.oracle_cfg_from_args(args)
    orc = OracleTPE(cfg, is_train=True, seed=17, step=5, dropout=True)
    sp = model.last['specaug_params'].cpu().numpy()
    ref = orc.loss_and_grads({k: v.astype(np.float64) for k, v in P.items()}, oracle_batch(batch), specaug_params=sp)
    assert abs(out4[0] - ref['loss']) < 2e-5 * abs(ref['loss'])
    compare_grads(model, ref, 1e-4)


@pytest.mark.parametrize('case', ['lut', 'costt'])
def test_bf16_within_stated_tolerance(case):
    """bf16 operands / fp32 accumulate and statistics, against the fp64 oracle at the tiny test width
    (d=16, where one bf16 rounding is a large fraction of a row): loss within 2e-2 relative, every
    gradient tensor within 0.15 of its largest entry.  The width-768 figure is measured by bench.py."""
    args = ge.tiny_args(compute_dtype='bf16', **CASES[case])
    model = make_model(args)
    P = perturb(model)
    P = {k: (torch.from_numpy(v).bfloat16().float().numpy() if v.ndim > 1 and 'embedding' not in k and 'conv' not in k else v) for k, v in P.items()}
    model.store.load_params(P)
    batch = make_batch(B=3, Traw=40, V=11, L=6, U=3, Lb=5, bert_dim=16, dim_raw=4, ragged=True, seed=2)
    out4 = model.forward_backward(batch).cpu().numpy()
    cfg = ge.oracle_cfg_from_args(args).zero_dropout()
    ref = OracleTPE(cfg, is_train=True, dropout=False).loss_and_grads({k: v.astype(np.float64) for k, v in P.items()}, oracle_batch(batch))
    assert abs(out4[0] - ref['loss']) < 2e-2 * abs(ref['loss'])
    if case == 'lut':   # with shrink, bf16 rounding may flip an arg-max and change which frames are kept
        compare_grads(model, ref, 0.15)


def test_fp32_greedy_ids_bit_exact_and_ctc_preds():
    args = ge.tiny_args(compute_dtype='fp32')
    model = make_model(args, is_train=False)
    P = perturb(model, seed=4)
    batch = make_batch(B=3, Traw=30, V=11, L=5, U=3, dim_raw=4, ragged=True, seed=9)
    preds, preds_am = model.infer(batch['feature'], batch['len_fea'])
    cfg = ge.oracle_cfg_from_args(args).zero_dropout()
    ob = oracle_batch(batch)
    rp, rpam, rlen = OracleTPE(cfg, is_train=False).infer({k: v.astype(np.float64) for k, v in P.items()}, ob['feature'], ob['len_fea'])
    assert np.array_equal(preds, rp), (preds, rp)
    assert np.array_equal(preds_am, rpam)
    assert np.array_equal(model.last['len_decoded'], rlen)


@pytest.mark.parametrize('case,beam,lpw,max_len,seed,boost', [
    ('lut', 4, 0.0, 12, 4, 0.2), ('lut', 4, 0.2, 12, 5, 0.2), ('costt', 3, 0.2, 30, 6, 0.2), ('lut', 2, 0.6, 9, 7, 0.2), ('lut', 8, 0.0, 16, 8, 0.2),
    ('lut', 4, 0.2, 30, 9, 0.16), ('costt', 4, 0.0, 30, 10, 0.17), ('lut', 3, 1.0, 30, 11, 0.15), ('lut', 4, 0.2, 30, 12, 0.0)])
def test_fp32_beam_search_ids_bit_exact(case, beam, lpw, max_len, seed, boost):
    """beam_decode_rerank (transformer_decoder.py:248-431) through TransformerTPE.infer with beam_size > 1: sorted
    hypotheses, best ids and lengths equal the oracle's bit for bit in fp32 mode; scores to 1e-5.  max_len 9 / 16 stop
    on and off the host's 8-step check; 30 lets every hypothesis reach <eos> when it can."""
    args = ge.tiny_args(compute_dtype='fp32', **CASES[case])
    args.beam_size, args.length_penalty_weight, args.max_len = beam, lpw, max_len
    model = make_model(args, is_train=False)
    P = perturb(model, seed=seed)
    rng = np.random.default_rng(seed)
    # decoder_fc has no bias: a common offset on the decoder's LayerNorm biases times a constant added to the <eos> column
    # raises the <eos> logit by about 0.5 * 16 * boost, enough for hypotheses to finish at different steps
    fc = [k for k in P if k.endswith('decoder_fc/kernel')][0]
    for k in P:
        if '/decoder/block_' in k and k.endswith('layer_norm_bias'):
            P[k] = (P[k] + 0.5).astype(np.float32)
    P[fc] = (P[fc] + 0.3 * rng.standard_normal(P[fc].shape)).astype(np.float32)
    P[fc][:, args.eos_idx] += boost
    model.store.load_params(P)
    batch = make_batch(B=3, Traw=30, V=11, L=5, U=3, dim_raw=4, ragged=True, seed=seed)
    preds, preds_am = model.infer(batch['feature'], batch['len_fea'])
    cfg = ge.oracle_cfg_from_args(args).zero_dropout()
    ob = oracle_batch(batch)
    rp, rpam, (r_sorted, r_scores, _, r_len) = OracleTPE(cfg, is_train=False).infer_beam(
        {k: v.astype(np.float64) for k, v in P.items()}, ob['feature'], ob['len_fea'], beam, lpw)
    (p_sorted, scores, _), = [model.last['beams']]
    assert preds.shape == rp.shape, (preds.shape, rp.shape)
    assert np.array_equal(p_sorted, r_sorted), (p_sorted, r_sorted)
    assert np.array_equal(preds, rp)
    assert np.array_equal(preds_am, rpam)
    assert np.array_equal(model.last['len_decoded'], r_len)
    live = r_scores > -1e9
    assert np.array_equal(scores > -1e9, live)
    assert np.abs(scores - r_scores)[live].max() < 1e-5 * max(np.abs(r_scores[live]).max(), 1.0)
    print('beam case', case, beam, lpw, 'steps', rp.shape[1], 'of', max_len, 'len_decoded', r_len.tolist(), 'best', rp.tolist())


@pytest.mark.parametrize('bucket_mb,overlap', [(None, '1'), (0.004, '1'), (0.004, '0')])
def test_train_steps_follow_oracle_adam_and_lr(bucket_mb, overlap, monkeypatch):
    """bucket_mb 0.004: ~40 gradient buckets of 1024 elements, each updated by its own Adam launch on the side stream
    while backward is still running (overlap '1') or all at once after it ('0')."""
    monkeypatch.setenv('NEURST_OVERLAP_ADAM', overlap)
    args = ge.tiny_args(compute_dtype='fp32')
    args.grad_bucket_mb = bucket_mb
    model = make_model(args, step=0)
    if bucket_mb:
        assert len(model.bucketer.bounds) > 8
    P = {k: v.astype(np.float64) for k, v in perturb(model).items()}
    M = {k: np.zeros_like(v) for k, v in P.items()}
    V = {k: np.zeros_like(v) for k, v in P.items()}
    cfg = ge.oracle_cfg_from_args(args).zero_dropout()
    orc = OracleTPE(cfg, is_train=True, dropout=False)
    batch = make_batch(B=2, Traw=24, V=11, L=5, U=3, Lb=4, bert_dim=16, dim_raw=4, seed=1)
    ob = oracle_batch(batch)
    for step in range(4):
        lr_before = model.learning_rate
        loss, shape_batch, l_ast, l_ctc, l_bert = model.train_step(batch)
        out, lr = oracle_train_step(orc, P, M, V, ob, step, args.warmup_steps, args.peak, args.decay_steps)
        assert abs(lr_before - lr) < 1e-9 and model.global_step.value == step + 1
        assert abs(loss - out['loss']) < 3e-5 * abs(out['loss']), (step, loss, out['loss'])
        assert shape_batch == ob['feature'].shape
    Pd = model.store.export()
    for k in P:
        # entries whose true gradient is ~0 (e.g. the key bias: softmax is shift invariant) move by
        # rounding noise / (sqrt(v)+1e-9) under Adam in ANY fp32 implementation; compare the others
        sel = np.sqrt(V[k]) > 1e-5
        if sel.any():
            assert np.abs(Pd[k] - P[k])[sel].max() < 2e-4 * max(np.abs(P[k]).max(), 1e-3), k


def test_l2_penalty_and_global_norm_clip_follow_the_oracle():
    """handle_gradients (gradientTools.py:40-67): L2 on non-bias variables, then tf.clip_by_global_norm - three train steps
    against the oracle's restatement, with a clip small enough to bind."""
    args = ge.tiny_args(compute_dtype='fp32')
    args.lambda_l2, args.grad_clip_global_norm = 0.05, 0.25
    args.grad_clip_value, args.grad_clip_norm = 3.0, 2.0        # only touch *lstm* variables: accepted, no effect
    model = make_model(args, step=0)
    P = {k: v.astype(np.float64) for k, v in perturb(model).items()}
    M = {k: np.zeros_like(v) for k, v in P.items()}
    V = {k: np.zeros_like(v) for k, v in P.items()}
    cfg = ge.oracle_cfg_from_args(args).zero_dropout()
    orc = OracleTPE(cfg, is_train=True, dropout=False)
    batch = make_batch(B=2, Traw=24, V=11, L=5, U=3, Lb=4, bert_dim=16, dim_raw=4, seed=1)
    ob = oracle_batch(batch)
    out0 = orc.loss_and_grads(P, ob, None)
    gn = np.sqrt(sum(float((g ** 2).sum()) for g in out0['grads'].values()))
    assert gn > 2 * args.grad_clip_global_norm                   # the clip binds
    for step in range(3):
        loss = model.train_step(batch)[0]
        out, lr = oracle_train_step(orc, P, M, V, ob, step, args.warmup_steps, args.peak, args.decay_steps, lambda_l2=0.05,
                                    grad_clip_global_norm=0.25)
        assert abs(loss - out['loss']) < 3e-5 * abs(out['loss']), (step, loss, out['loss'])
    Pd, Md = model.store.export(), model.store.export(model.store.m)
    for k in P:
        assert np.abs(Md[k] - M[k]).max() < 2e-4 * max(np.abs(M[k]).max(), 1e-6), k      # first moments: the handled gradients themselves
        sel = np.sqrt(V[k]) > 1e-5
        if sel.any():
            assert np.abs(Pd[k] - P[k])[sel].max() < 2e-4 * max(np.abs(P[k]).max(), 1e-3), k


def test_frozen_scope_is_filtered_after_handle_gradients():
    """frozen_scope (TransformerTPE.py:170-183) with the L2 penalty and the global-norm clip on: the frozen variables and
    their Adam slots (pre-loaded with non-zero moments, as after a restore) stay BIT-identical over three steps, their
    gradients still count in the clip's norm, and every other variable follows the oracle."""
    args = ge.tiny_args(compute_dtype='fp32')
    args.lambda_l2, args.grad_clip_global_norm = 0.05, 0.25
    args.frozen_scope = 'TPEncoder,embedding'
    model = make_model(args, step=0)
    assert model.frozen == ['TPEncoder', 'embedding']
    P = {k: v.astype(np.float64) for k, v in perturb(model).items()}
    rng = np.random.default_rng(5)
    M = {k: 0.01 * rng.standard_normal(v.shape) for k, v in P.items()}
    V = {k: 1e-4 * rng.random(v.shape) for k, v in P.items()}
    model.store.load_slots(M, V)
    frozen = [k for k in P if 'TPEncoder' in k or 'embedding' in k]
    assert frozen and len(frozen) < len(P)
    P0, M0, V0 = model.store.export(), model.store.export(model.store.m), model.store.export(model.store.v)
    M = {k: M0[k].astype(np.float64) for k in P}                 # the fp32 values the device holds
    V = {k: V0[k].astype(np.float64) for k in P}
    cfg = ge.oracle_cfg_from_args(args).zero_dropout()
    orc = OracleTPE(cfg, is_train=True, dropout=False)
    batch = make_batch(B=2, Traw=24, V=11, L=5, U=3, Lb=4, bert_dim=16, dim_raw=4, seed=1)
    ob = oracle_batch(batch)
    for step in range(3):
        loss = model.train_step(batch)[0]
        out, lr = oracle_train_step(orc, P, M, V, ob, step, args.warmup_steps, args.peak, args.decay_steps, lambda_l2=0.05,
                                    grad_clip_global_norm=0.25, frozen_scope=args.frozen_scope)
        assert abs(loss - out['loss']) < 3e-5 * abs(out['loss']), (step, loss, out['loss'])
        if step == 0:                                            # the frozen gradients are a real share of the norm
            gf = np.sqrt(sum(float((out['grads'][k] ** 2).sum()) for k in frozen))
            ga = np.sqrt(sum(float((g ** 2).sum()) for g in out['grads'].values()))
            assert gf > 0.05 * ga
    Pd, Md, Vd = model.store.export(), model.store.export(model.store.m), model.store.export(model.store.v)
    for k in frozen:
        assert np.array_equal(Pd[k], P0[k]) and np.array_equal(Md[k], M0[k]) and np.array_equal(Vd[k], V0[k]), k
    for k in P:
        if k in frozen:
            continue
        assert np.abs(Md[k] - M[k]).max() < 2e-4 * max(np.abs(M[k]).max(), 1e-6), k
        assert np.abs(Pd[k] - P[k]).max() < 2e-4 * max(np.abs(P[k]).max(), 1e-3), k
    assert model.graph.produced == set(P)                        # every variable of the LUT graph receives a gradient


def test_train_and_infer_models_share_variables():
    args = ge.tiny_args(compute_dtype='fp32')
    train = make_model(args)
    infer = make_model(args, is_train=False)
    assert train.store is infer.store
    batch = make_batch(B=2, Traw=24, V=11, L=5, U=3, Lb=4, bert_dim=16, dim_raw=4, seed=1)
    l0 = train.train_step(batch)[0]
    for _ in range(30):
        l1 = train.train_step(batch)[0]
    assert l1 < l0
    preds, preds_am = infer.infer(batch['feature'], batch['len_fea'])
    assert preds.shape[0] == 2


def test_full_width_lut_block_shapes_bf16_runs_and_is_finite():
    """LUT widths (d=768, H=8, dh=96, KD dh=192, Din=480) with one block each and a short batch:
    exercises the vector paths of every kernel at the template's dimensions."""
    from neurst_amd.st.layers.arguments import build_args, load_config
    import os
    cfg = load_config(os.path.join(ge.ROOT, 'configs', 'lut.yaml'))
    for part in ('encoder', 'encoder2', 'decoder'):
        cfg['model'][part]['num_blocks'] = 1
    cfg['beam_size'] = 1
    args = build_args(cfg, dim_output=1000, compute_dtype='bf16')
    model = make_model(args)
    batch = make_batch(B=4, Traw=300, V=1000, L=12, U=10, Lb=9, bert_dim=768, ragged=True)
    vals = [model.train_step(batch)[0] for _ in range(3)]
    assert all(np.isfinite(v) for v in vals), vals
    g = model.store.export_grads()
    assert all(np.isfinite(v).all() for v in g.values())


def test_full_width_bf16_against_fp32_mode():
    """The bf16 tolerance at the template's width (d=768, 2+2+2 blocks, V=1000, dropout 0): the bf16 path
    (bf16 operands and activations, fp32 accumulate / statistics / master weights) against the fp32 parity path on the
    same weights and batch.  Stated tolerance: total loss within 5e-3 relative, every gradient tensor with relative L2
    error < 8e-2, cosine similarity > 0.995 and no entry further than 0.2 of the tensor's largest entry from fp32."""
    from neurst_amd.st.layers.arguments import build_args, load_config
    import os
    out = {}
    P = None
    batch = make_batch(B=4, Traw=240, V=1000, L=12, U=10, Lb=9, bert_dim=768, ragged=True, seed=21)
    for mode in ('fp32', 'bf16'):
        cfg = load_config(os.path.join(ge.ROOT, 'configs', 'lut.yaml'))
        for part in ('encoder', 'encoder2', 'decoder'):
            cfg['model'][part]['num_blocks'] = 2
            cfg['model'][part]['attention_dropout_rate'] = 0.0
            cfg['model'][part]['residual_dropout_rate'] = 0.0
        cfg['model']['kd']['attention_dropout_rate'] = 0.0
        cfg['beam_size'] = 1
        args = build_args(cfg, dim_output=1000, compute_dtype=mode)
        model = make_model(args)
        if P is None:
            P = perturb(model, seed=3)
        else:
            model.store.load_params(P)
        out4 = model.forward_backward(batch).cpu().numpy()
        out[mode] = (out4, model.store.export_grads())
    l32, g32 = out['fp32']
    l16, g16 = out['bf16']
    assert abs(l16[0] - l32[0]) < 5e-3 * abs(l32[0]), (l16, l32)
    worst, worst_l2, worst_cos = 0.0, 0.0, 1.0
    for k, v in g32.items():
        if np.abs(v).max() < 1e-12:
            continue
        err = np.abs(g16[k] - v).max() / np.abs(v).max()
        l2 = float(np.linalg.norm(g16[k] - v) / np.linalg.norm(v))
        cos = float((g16[k] * v).sum() / (np.linalg.norm(g16[k]) * np.linalg.norm(v) + 1e-30))
        worst, worst_l2, worst_cos = max(worst, err), max(worst_l2, l2), min(worst_cos, cos)
        assert err < 0.2 and l2 < 0.08 and cos > 0.995, (k, err, l2, cos)
    print('bf16 vs fp32 at d=768: loss rel %.2e, worst grad: max-entry err %.3f of max, relative L2 %.3f, cosine %.5f'
          % (abs(l16[0] - l32[0]) / abs(l32[0]), worst, worst_l2, worst_cos))


def test_template_width_bf16_against_the_oracle():
    """The bf16 product path against the fp64 ORACLE (not against the fp32 kernels) at the template's width: d = 768, H = 8
    (dh = 96), KD heads of 192, Din = 480, one block each, V = 1000, B = 2 utterances of ~40 encoder frames, dropout 0.
    Stated tolerance: total loss within 5e-3 relative; every gradient tensor with relative L2 error < 8e-2, cosine > 0.995
    and no entry further than 0.3 of the tensor's largest entry from the oracle's (measured on MI355X: loss 1e-4, worst
    tensor relative L2 0.062 / cosine 0.998 / single entry 0.22 of max - with two utterances a weight-gradient entry is
    a sum of ~80 bf16-rounded products, so single entries scatter more than the tensor as a whole)."""
    from neurst_amd.st.layers.arguments import build_args, load_config
    import os
    cfg = load_config(os.path.join(ge.ROOT, 'configs', 'lut.yaml'))
    for part in ('encoder', 'encoder2', 'decoder'):
        cfg['model'][part]['num_blocks'] = 1
        cfg['model'][part]['attention_dropout_rate'] = 0.0
        cfg['model'][part]['residual_dropout_rate'] = 0.0
    cfg['model']['kd']['attention_dropout_rate'] = 0.0
    cfg['beam_size'] = 1
    args = build_args(cfg, dim_output=1000, compute_dtype='bf16')
    model = make_model(args)
    P = perturb(model, seed=4)
    batch = make_batch(B=2, Traw=120, V=1000, L=8, U=6, Lb=7, bert_dim=768, ragged=True, seed=31)
    out4 = model.forward_backward(batch).cpu().numpy()
    g16 = model.store.export_grads()
    ocfg = ge.oracle_cfg_from_args(args).zero_dropout()
    ref = OracleTPE(ocfg, is_train=True, dropout=False).loss_and_grads({k: v.astype(np.float64) for k, v in P.items()}, oracle_batch(batch))
    assert abs(out4[0] - ref['loss']) < 5e-3 * abs(ref['loss']), (out4, ref['loss'])
    worst, worst_l2, worst_cos = 0.0, 0.0, 1.0
    for k, v in ref['grads'].items():
        if v.size == 0 or np.abs(v).max() < 1e-12:
            continue
        err = np.abs(g16[k] - v).max() / np.abs(v).max()
        l2 = float(np.linalg.norm(g16[k] - v) / np.linalg.norm(v))
        cos = float((g16[k] * v).sum() / (np.linalg.norm(g16[k]) * np.linalg.norm(v) + 1e-30))
        worst, worst_l2, worst_cos = max(worst, err), max(worst_l2, l2), min(worst_cos, cos)
        assert err < 0.3 and l2 < 0.08 and cos > 0.995, (k, err, l2, cos)
    print('bf16 vs oracle at d=768: loss rel %.2e, worst grad: max-entry err %.3f of max, relative L2 %.3f, cosine %.5f'
          % (abs(out4[0] - ref['loss']) / abs(ref['loss']), worst, worst_l2, worst_cos))


def test_costt_speculative_shrink_bound_matches_the_read_back_path(monkeypatch):
    """shrink_layer sized from earlier steps' maxima (no mid-step device -> host read) against the reference's sizing (read
    the longest kept sequence back, blocks.py:26-36), four fp32 train steps over batches of growing length: the same losses
    (1e-5 relative) and parameters (well-conditioned entries, 1e-4).  The third batch keeps more frames than the bound
    guessed from the first two, so its Adam update must have been vetoed on the device and the batch repeated at full
    length exactly once - global_step and adam_t still count four steps."""
    batches = [make_batch(B=3, Traw=Traw, V=11, L=6, U=3, dim_raw=4, ragged=True, seed=40 + i) for i, Traw in enumerate((24, 24, 400, 60))]
    got = {}
    for spec in ('0', '1'):
        monkeypatch.setenv('NEURST_SHRINK_SPEC', spec)
        args = ge.tiny_args(compute_dtype='fp32', **CASES['costt'])
        model = make_model(args)
        if spec == '1':
            assert model.graph.shrink_spec is not None
            model.graph.shrink_spec.quantum = 4          # tiny shapes: keep the guessed bound close to the observed maxima
            model.graph.shrink_spec.slack = 1
            P = got['0'][2]
            model.store.load_params(P)
        else:
            assert model.graph.shrink_spec is None
            P = perturb(model, seed=6)
        losses, kept = [], []
        for b in batches:
            losses.append(model.train_step(b)[0])
            kept.append(int(model.last['len_encoded'].max().item()))
        torch.cuda.synchronize()
        got[spec] = (losses, model.store.params.clone(), P, kept, model.global_step.value, model.store.adam_t, model.last.get('shrink_replays', 0))
    l0, p0, _, k0, gs0, t0, _ = got['0']
    l1, p1, _, k1, gs1, t1, replays = got['1']
    assert k0 == k1 and k0[2] > 1.25 * max(k0[:2]) + 4, k0       # the third batch really outgrows the guessed bound
    assert replays == 1, replays
    assert (gs0, t0) == (gs1, t1) and t1 - 0 >= 4
    assert np.allclose(l1, l0, rtol=1e-5), (l1, l0)
    sel = p0.abs() > 1e-4 * p0.abs().max()
    assert torch.allclose(p1[sel], p0[sel], rtol=1e-4, atol=1e-6)


RCCL_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as ge
from neurst_amd.synthetic import make_batch
from neurst_amd.st.models.TransformerTPE import GlobalStep
torch.cuda.set_device(0)
os.environ['NEURST_FORCE_BUCKETS'] = '1'
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%%s' %% sys.argv[1], rank=0, world_size=1, device_id=torch.device('cuda', 0))
batch = make_batch(B=3, Traw=40, V=11, L=6, U=3, Lb=5, bert_dim=16, dim_raw=4, ragged=True, seed=1)
results = {}
for mode in ('plain', 'buckets', 'buckets+adam', 'buckets+tail'):
    os.environ['NEURST_OVERLAP_ADAM'] = {'buckets+adam': 'backward', 'buckets+tail': 'tail'}.get(mode, 'off')
    args = ge.tiny_args(compute_dtype='fp32')
    args.grad_bucket_mb = 0.004
    model = args.Model(GlobalStep(0), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
    if mode == 'plain':
        model.bucketer.reduce = False
    else:
        assert model.bucketer.reduce and model.bucketer.comm_stream is not None and len(model.bucketer.bounds) > 8
    losses = [model.train_step(batch)[0]]
    g1 = model.store.grads.clone()             # reduced in place, untouched by Adam
    p1 = model.store.params.clone()
    losses += [model.train_step(batch)[0] for _ in range(2)]
    torch.cuda.synchronize()
    results[mode] = (losses, g1, p1)
for mode in ('buckets', 'buckets+adam', 'buckets+tail'):
    # the LN column sums are fp32 atomics (order varies run to run) and Adam divides by sqrt(v) + 1e-9, so parameters whose
    # gradient is rounding noise may move differently: compare the gradients, the well-conditioned parameters and the losses
    assert np.allclose(results[mode][0], results['plain'][0], rtol=1e-5), (mode, results[mode][0], results['plain'][0])
    ga, gb = results[mode][1], results['plain'][1]
    assert (ga - gb).abs().max().item() < 1e-6 * gb.abs().max().item(), mode
    sel = gb.abs() > 1e-4 * gb.abs().max()
    assert torch.allclose(results[mode][2][sel], results['plain'][2][sel], rtol=1e-4, atol=1e-6), mode
# the template's width with the second streams busy (weight gradients, CTC): buckets of 4 MB reduced while backward runs
from neurst_amd.st.layers.arguments import build_args, load_config
wide = {}
for mode in ('plain', 'buckets'):
    cfg = load_config(os.path.join(ge.ROOT, 'configs', 'lut.yaml'))
    for part in ('encoder', 'encoder2', 'decoder'):
        cfg['model'][part]['num_blocks'] = 1
    cfg['beam_size'], cfg['grad_bucket_mb'] = 1, 4
    args = build_args(cfg, dim_output=1000, compute_dtype='bf16')
    model = args.Model(GlobalStep(1000), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
    model.bucketer.reduce = mode == 'buckets'
    os.environ.pop('NEURST_OVERLAP_ADAM', None)      # default: one Adam launch after the gradient mean
    b2 = make_batch(B=8, Traw=600, V=1000, L=12, U=10, Lb=9, bert_dim=768, ragged=True, seed=4)
    wide[mode] = [model.train_step(b2)[0] for _ in range(4)]
assert np.allclose(wide['buckets'], wide['plain'], rtol=2e-3), wide
dist.destroy_process_group()
print('rccl one-rank ok')
'''


def test_bucketed_allreduce_plumbing_over_rccl_one_rank(tmp_path):
    """The N > 1 code path on the real backend (RCCL collectives launched from the communication stream during backward,
    async handles waited on streams, optional in-backward Adam), forced on with ONE rank: an all-reduce over one rank is the
    identity, so three train steps must match the plain path.  (More than one rank per GPU is not possible with RCCL; the
    multi-rank arithmetic is covered by the 2-rank gloo test on CPU.)"""
    import os
    import subprocess
    import sys
    script = tmp_path / 'worker.py'
    script.write_text(RCCL_WORKER % ge.ROOT)
    p = subprocess.run([sys.executable, str(script), str(29600 + os.getpid() % 2000)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=150)
    out = p.stdout.decode()
    assert p.returncode == 0 and 'rccl one-rank ok' in out, out[-3000:]


TWO_RANK_WORKER = r"""
import os, sys, faulthandler
faulthandler.dump_traceback_later(100, exit=True)
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as ge
from neurst_amd.synthetic import make_batch
from neurst_amd.st.models.TransformerTPE import GlobalStep
rank, port, outdir, loss_type, wire = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5]
torch.cuda.set_device(0)
# two ranks on ONE GPU: RCCL refuses that, so the process group is gloo; the bucketer then stages every bucket through
# pinned host memory (gradientTools.GradBucketer.host_stage)
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + port, rank=rank, world_size=2)
args = ge.tiny_args(loss_type=loss_type, compute_dtype='fp32', shrink=(loss_type == 'ce_ctc'))
args.grad_bucket_mb = 0.004                      # about 1000 floats per bucket: more than 8 buckets in flight during backward
model = args.Model(GlobalStep(3), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
assert model.bucketer.reduce and model.bucketer.host_stage and model.bucketer.world == 2 and len(model.bucketer.bounds) > 8
model.bucketer.bf16 = wire == 'bf16'             # buckets cast to bf16 on the communication stream, summed in bf16, widened back
full = make_batch(B=4, Traw=30, V=11, L=6, U=3, Lb=5, bert_dim=16 if loss_type == 'ctc_kd_ce' else None, dim_raw=4, ragged=True, seed=5)
half = {k: np.ascontiguousarray(v[2 * rank:2 * rank + 2]) for k, v in full.items()}      # tf.split after padding (lstmModel.py:342-350)
P0 = model.store.export()
out4 = model.forward_backward(half).cpu().numpy()
launched_early = sum(model.bucketer.launched)
from neurst_amd.st.tools.tftools.gradientTools import average_gradients
scale = average_gradients(model.bucketer)          # finish(): every bucket reduced (sum over the two ranks)
torch.cuda.synchronize()
assert scale == 0.5 and launched_early > 0
g = {k: v * scale for k, v in model.store.export_grads().items()}
np.savez(os.path.join(outdir, 'rank%%d.npz' %% rank), loss=out4, **{'g/' + k: v for k, v in g.items()}, **{'p/' + k: v for k, v in P0.items()})
# two more whole steps (reduce + Adam): both ranks must hold identical parameters afterwards
for _ in range(2):
    model.train_step(half)
torch.cuda.synchronize()
np.savez(os.path.join(outdir, 'params%%d.npz' %% rank), **model.store.export())
dist.destroy_process_group()
print('two-rank worker ok', rank, flush=True)
"""


@pytest.mark.parametrize('loss_type,wire', [('ctc_kd_ce', 'f32'), ('ce_ctc', 'f32'), ('ctc_kd_ce', 'bf16')])
def test_two_ranks_on_one_gpu_reduce_to_the_full_batch_gradient(tmp_path, loss_type, wire):
    """a19 with world = 2 on the device (gradientTools.py:7-37, lstmModel.py:342-350): two fresh processes share cuda:0, each
    runs the real train step (fp32 mode) on its half of a padded batch with the bucketed all-reduce launched during backward;
    the reduced, averaged gradient must equal the oracle's full-batch gradient (the reference's mean over towers), and after
    two further steps both ranks hold the same parameters.  wire = bf16: the buckets travel as bf16 (GradBucketer.bf16) -
    both addends and their sum are rounded to 8 significant bits, so the stated tolerance against the oracle is 2^-7 of
    each tensor's largest entry instead of 5e-5; the ranks still agree bit for bit."""
    import os
    import subprocess
    import sys
    from oracle import ops as oops
    from oracle.model import OracleTPE
    script = tmp_path / 'worker.py'
    script.write_text(TWO_RANK_WORKER % ge.ROOT)
    port = str(29800 + os.getpid() % 1500)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port, str(tmp_path), loss_type, wire], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=140)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and 'two-rank worker ok' in o, 'rank %d:\n%s' % (r, o[-3000:])
    r0, r1 = (np.load(str(tmp_path / ('rank%d.npz' % r))) for r in range(2))
    keys = [k[2:] for k in r0.files if k.startswith('g/')]
    for k in keys:                                   # every rank holds the same reduced gradient and started from the same weights
        assert np.array_equal(r0['g/' + k], r1['g/' + k]), k
        assert np.array_equal(r0['p/' + k], r1['p/' + k]), k
    # oracle: the full batch of 4 in one tower
    from neurst_amd.synthetic import make_batch
    args = ge.tiny_args(loss_type=loss_type, compute_dtype='fp32', shrink=(loss_type == 'ce_ctc'))
    full = make_batch(B=4, Traw=30, V=11, L=6, U=3, Lb=5, bert_dim=16 if loss_type == 'ctc_kd_ce' else None, dim_raw=4, ragged=True, seed=5)
    cfg = ge.oracle_cfg_from_args(args).zero_dropout()
    feat, lens = oops.splice_downsample_batch(full['feature'], full['len_fea'], 5, 3)
    P = {k: r0['p/' + k] for k in keys}
    ref = OracleTPE(cfg, is_train=True, dropout=False).loss_and_grads(P, dict(full, feature=feat, len_fea=lens))
    mean_loss = 0.5 * (float(r0['loss'][0]) + float(r1['loss'][0]))
    if loss_type == 'ce_ctc':                        # without the KD term the loss is a plain mean over utterances
        assert abs(mean_loss - ref['loss']) <= 1e-5 * abs(ref['loss']), (mean_loss, ref['loss'])
    for k, v in ref['grads'].items():
        if loss_type == 'ctc_kd_ce' and v.size == 0:
            continue
        err = np.abs(r0['g/' + k] - v).max()
        tol = 5e-5 if wire == 'f32' else 2.0 ** -7
        assert err <= tol * max(np.abs(v).max(), 1e-6) + 1e-7, (k, err, np.abs(v).max())
    if wire == 'bf16':                               # ... and the bf16 wire really was used: some entry differs from the fp32 exchange's
        assert any(np.abs(r0['g/' + k] - v).max() > 5e-5 * max(np.abs(v).max(), 1e-6) + 1e-7 for k, v in ref['grads'].items() if v.size)
    pa, pb = (np.load(str(tmp_path / ('params%d.npz' % r))) for r in range(2))
    for k in pa.files:
        assert np.array_equal(pa[k], pb[k]), k


def test_bench_prints_exactly_one_json_line_with_the_contract_keys():
    """bench.py's contract: stdout is ONE JSON line (library banners go to stderr) carrying the driver's keys and the
    roofline object; under torch.distributed.run with one rank the same holds (RCCL prints a version banner to stdout)"""
    import json
    import os
    import subprocess
    import sys
    root = ge.ROOT
    base = [os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--no-cpu-baseline']
    runs = [[sys.executable] + base,
            [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
             '--master-port', '29713'] + base,
            [sys.executable] + base + ['--config', os.path.join(root, 'configs', 'costt.yaml')]]      # BASELINE config 3
    for cmd in runs:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.split('\n') if l.strip()]
        assert len(lines) == 1, out.stdout[:2000]
        d = json.loads(lines[0])
        for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
                  'data', 'config', 'roofline'):
            assert k in d, k
        assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['vs_baseline'] is None
        assert d['higher_is_better'] is True and d['dtype'] == 'bf16' and 'workload' in d['config'] and 'model' not in d['config']
        r = d['roofline']
        for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
            assert k in r, k
        assert r['bound'] in ('hbm', 'mfma') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
        assert abs(d['value'] - 32 * 500 / (d['ms_per_step'] / 1e3)) < 1e-3 * d['value']
        assert ('COSTT' in d['config']['workload']) == ('costt.yaml' in cmd[-1])
        assert 0 < r['step_frac'] < 1 and 'traffic_stale' in r
        # every launch class is timed, not only the GEMMs: the dominant kernel is chosen over the whole step
        other = r['other_ms_per_step']
        assert any('attention backward' in k for k in other) and any('layernorm' in k for k in other) and 'adam' in other
        assert any('ctc loss' in k for k in other) and r['kernel'] in list(other) + list(r['gemm_ms_per_step'])
        if 'costt.yaml' not in cmd[-1]:               # BASELINE config 3 rides along in the LUT line
            assert d['secondary']['costt']['ms_per_step'] > 0 and 'COSTT' in d['secondary']['costt']['workload']


def test_two_models_in_one_process_share_their_side_streams():
    """HIP gives a process a handful of hardware queues: a second model must not bring a second set of side streams
    (graph.shared_stream; with one set per model bench.py's COSTT step behind the LUT step ran 15.6 instead of 9.3 ms)."""
    from neurst_amd.st.models import graph as G
    batch = make_batch(B=2, Traw=24, V=11, L=5, U=3, Lb=4, bert_dim=16, dim_raw=4, ragged=True)
    models = []
    for lt in ('ctc_kd_ce', 'ce_ctc'):
        args = ge.tiny_args(loss_type=lt, compute_dtype='bf16')
        m = make_model(args)
        m.train_step(batch)
        models.append(m)
    a, b = models
    assert a.graph.wgrad_stream is not None and a.graph.wgrad_stream is b.graph.wgrad_stream
    assert a._ctc_stream is not None and a._ctc_stream is b._ctc_stream
    assert G.shared_stream('wgrad', a.device) is a.graph.wgrad_stream
    assert np.isfinite(a.train_step(batch)[0]) and np.isfinite(b.train_step(batch)[0])     # interleaved use of the shared streams
