"""-m gpu: seeded random walk over the whole TransformerTPE train step (st/models/TransformerTPE.py build_single_graph,
reference /root/reference/st/models/TransformerTPE.py:195-484) through the host mirror and the C ABI, against the fp64
numpy oracle with identical weights: both loss mixes (LUT ctc_kd_ce, COSTT ce_ctc + shrink), widths, head counts, block
counts per part, vocabulary, batch, ragged frame / label / transcript lengths, and dropout (the oracle draws the same
decisions from the same hash) are drawn independently.  fp32 parity mode, the north-star tolerance: losses 1e-5 (2e-5 with
dropout), every gradient tensor 1e-4 of its largest entry (floor 1e-3 on tensors whose exact gradient is zero: one visible key)."""
import copy

import os

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from oracle import ops as O
from oracle.model import OracleTPE
from neurst_amd.synthetic import make_batch

pytestmark = pytest.mark.gpu
FUZZ_SCALE = int(os.environ.get('NEURST_FUZZ_SCALE', '1'))     # more seeds for a longer hunt (not the committed suite)


def _args(rng, case, dropout):
    from neurst_amd.st.layers.arguments import build_args
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]
    heads = pick([1, 2, 4])
    d = heads * pick([4, 8, 16])
    V = pick([7, 11, 40])
    rate = 0.1 if dropout else 0.0

    def blk():
        return dict(num_blocks=int(rng.integers(1, 4)), num_heads=heads, num_cell_units=d, attention_dropout_rate=rate,
                    residual_dropout_rate=rate)

    kd_heads = pick([1, 2])
    cfg = {
        'data': {'dim_raw_input': 4, 'num_context': 5, 'downsample': 3, 'add_delta': False, 'feature_channel': 1},
        'model': {
            'encoder': dict(blk(), type='transformer_encoder'),
            'encoder2': dict(blk(), type='transformer_tpeencoder'),
            'decoder': dict(blk(), type='transformer_decoder', size_embedding=d, label_smoothing_confidence=pick([0.9, 1.0])),
            'decoder2': {'type': 'fc_decoder', 'ctc_merge_repeated': True, 'kd_factor': pick([0.5, 0.3])},
            'kd': {'feat_extract_type': 'self_attention', 'loss_type': 'mean_squared_error', 'kd_factor': 0.05,
                   'num_cell_units': d, 'num_heads': kd_heads, 'attention_dropout_rate': 0.0},
            'structure': 'transformerTpe', 'loss_type': 'ctc_kd_ce' if case == 'lut' else 'ce_ctc', 'use_semantic_encoder': True,
            'shrink_layer': case == 'costt',
        },
        'gpus': '0', 'optimizer': 'adam', 'warmup_steps': 10, 'peak': 1e-3, 'decay_steps': 100, 'beam_size': 1, 'max_len': 12,
        'lambda_l2': 0.0, 'grad_clip_value': 0.0, 'grad_clip_norm': 0.0, 'grad_clip_global_norm': 0.0,
    }
    return build_args(copy.deepcopy(cfg), dim_output=V, compute_dtype='fp32'), d, V


@pytest.mark.parametrize('seed', range(16 * FUZZ_SCALE))
def test_fp32_train_step_random_configs_match_oracle(seed, monkeypatch):
    from neurst_amd.st.models.TransformerTPE import GlobalStep
    if seed % 4 == 3:
        # COSTT with dropout: the decisions of a layer are a hash of (site key, flat element index), so behind shrink_layer they
        # depend on the padded length of the shrunk tensor.  The oracle pads to the longest kept sequence like the reference
        # (blocks.py:26-36); the product's default sizes it from earlier steps without a read-back (ShrinkSpeculation) - the same
        # values on every valid position without dropout (the seeds with seed % 4 == 1 run that path), other decisions of the
        # same distribution with it.  The oracle's decisions are comparable on the read-back path only.
        monkeypatch.setenv('NEURST_SHRINK_SPEC', '0')
    rng = np.random.default_rng(31000 + seed)
    case = 'lut' if seed % 2 == 0 else 'costt'
    dropout = seed % 4 >= 2
    args, d, V = _args(rng, case, dropout)
    args.seed = 100 + seed
    step = int(rng.integers(0, 9))
    model = args.Model(GlobalStep(step), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
    P = model.store.export()
    for k, v in P.items():                       # biases and LayerNorm parameters away from their initial 0 / 1
        if v.ndim == 1:
            P[k] = (v + 0.1 * rng.standard_normal(v.shape)).astype(np.float32)
    model.store.load_params(P)
    B, Traw = int(rng.integers(1, 6)), int(rng.integers(3, 120))
    L, U, Lb = int(rng.integers(1, 10)), int(rng.integers(1, 7)), int(rng.integers(1, 8))
    batch = make_batch(B=B, Traw=Traw, V=V, L=L, U=U, Lb=Lb, bert_dim=d, dim_raw=4, ragged=bool(rng.integers(0, 2)), seed=seed)
    what = dict(seed=seed, case=case, dropout=dropout, d=d, V=V, B=B, Traw=Traw, L=L, U=U, Lb=Lb, step=step,
                blocks=[args.model.encoder.num_blocks, args.model.encoder2.num_blocks, args.model.decoder.num_blocks])
    out4 = model.forward_backward(batch).cpu().numpy()
    cfg = ge.oracle_cfg_from_args(args)
    feat, lens = O.splice_downsample_batch(batch['feature'], batch['len_fea'], 5, 3)
    ob = dict(batch, feature=feat, len_fea=lens)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    if dropout:
        ref = OracleTPE(cfg, is_train=True, seed=args.seed, step=step, dropout=True).loss_and_grads(P64, ob)
    else:
        ref = OracleTPE(cfg.zero_dropout(), is_train=True, dropout=False).loss_and_grads(P64, ob)
    ltol = 2e-5 if dropout else 1e-5
    names = ('loss', 'loss_ast', 'loss_ctc') + (('loss_bert',) if case == 'lut' else ())
    for got, name in zip(out4, names):
        want = float(ref[name])
        if np.isposinf(want):     # an utterance without a valid CTC path (repeats need more frames than it has): +inf like tf.nn.ctc_loss
            assert np.isposinf(got), (name, out4, want, what)
        else:
            assert abs(got - want) < ltol * max(abs(want), 1e-3), (name, out4, want, what)
    assert np.array_equal(model.last['len_encoded'].cpu().numpy(), ref['len_enc']), what
    g = model.store.export_grads()
    for k, v in ref['grads'].items():
        err = np.abs(g[k] - v).max() / max(np.abs(v).max(), 1e-3)
        assert err < 1e-4, (k, err, what)


@pytest.mark.parametrize('seed', range(10 * FUZZ_SCALE))
def test_fp32_greedy_decoding_random_configs_bit_exact(seed):
    """cached greedy decoding (TransformerTPE.py:366-484, transformer_decoder.py:184-246) of random configurations: token ids,
    CTC predictions and decoded lengths equal the oracle's bit for bit in fp32 mode"""
    from neurst_amd.st.models.TransformerTPE import GlobalStep
    rng = np.random.default_rng(47000 + seed)
    case = 'lut' if seed % 2 == 0 else 'costt'
    args, d, V = _args(rng, case, False)
    args.max_len = int(rng.integers(3, 20))
    model = args.Model(GlobalStep(0), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=False, args=args)
    P = model.store.export()
    for k, v in P.items():
        if v.ndim == 1:
            P[k] = (v + 0.1 * rng.standard_normal(v.shape)).astype(np.float32)
    model.store.load_params(P)
    B, Traw = int(rng.integers(1, 6)), int(rng.integers(3, 120))
    batch = make_batch(B=B, Traw=Traw, V=V, L=3, U=2, dim_raw=4, ragged=bool(rng.integers(0, 2)), seed=seed)
    what = dict(seed=seed, case=case, d=d, V=V, B=B, Traw=Traw, max_len=args.max_len)
    preds, preds_am = model.infer(batch['feature'], batch['len_fea'])
    cfg = ge.oracle_cfg_from_args(args).zero_dropout()
    feat, lens = O.splice_downsample_batch(batch['feature'], batch['len_fea'], 5, 3)
    rp, rpam, rlen = OracleTPE(cfg, is_train=False).infer({k: v.astype(np.float64) for k, v in P.items()}, feat, lens)
    assert np.array_equal(preds, rp), (preds, rp, what)
    assert np.array_equal(preds_am, rpam), what
    assert np.array_equal(model.last['len_decoded'], rlen), what
