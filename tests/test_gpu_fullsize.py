"""-m gpu: the HIP path at BASELINE config 2's FULL sizes (B=32, T=500, d=768, H=8, dh=96, V=10000), where the numpy
oracle would take minutes: checked through size-independent properties of the domain instead -
rows of attention weights sum to one, a CTC gradient frame sums to zero, LayerNorm rows are standardised, GEMM is
linear, the forward pass is deterministic, and the mean of tower gradients equals the full-batch gradient (the
reference's data-parallel rule, gradientTools.py:7-37).  torch is used here only to CHECK device buffers."""
import os

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from neurst_amd.synthetic import make_batch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from neurst_amd import ops
    DEV = torch.device('cuda:0')

B, H, T, DH, D, V = 32, 8, 500, 96, 768, 10000


def test_attention_rows_sum_to_one_and_dropout_keeps_the_mean():
    g = torch.Generator(device='cpu').manual_seed(1)
    qkv = (torch.randn(B, T, 3 * D, generator=g) * 0.5).to(DEV).to(torch.bfloat16)
    qkv[:, :, 2 * D:] = 1.0                                  # V = ones: O = sum_j P_ij = 1 for every query
    lens = torch.from_numpy(np.random.default_rng(0).integers(T // 2, T + 1, size=B).astype(np.int32)).to(DEV)
    t = qkv.view(-1)
    o, lse = ops.mha_fwd(t, t[D:], t[2 * D:], 3 * D, 3 * D, 3 * D, B, H, T, T, DH, ops.MASK_KEYPAD, lens, 0.0, 7)
    assert torch.isfinite(lse).all()
    assert (o.float() - 1.0).abs().max().item() < 1e-2        # bf16 probabilities: 2^-8 relative each
    o2, lse2, kb = ops.mha_fwd(t, t[D:], t[2 * D:], 3 * D, 3 * D, 3 * D, B, H, T, T, DH, ops.MASK_KEYPAD, lens, 0.1, 7, keep_bits=True)
    assert abs(o2.float().mean().item() - 1.0) < 2e-3         # E[keep / (1 - rate)] = 1
    # the stored decisions are the recomputed ones: the two-kernel backward with and without them gives the same gradients
    # bit for bit, and the fused backward (which only runs on stored decisions) the same up to where it rounds to bf16
    from neurst_amd import lib
    C = lib.load()
    do = torch.randn_like(o2)
    outs = []
    try:
        for fused, bits in ((0, kb), (0, None), (1, kb)):
            assert C.neurst_set_option(b'mha_fused', fused) == 0
            dqkv = torch.zeros_like(qkv)
            dt = dqkv.view(-1)
            ops.mha_bwd(t, t[D:], t[2 * D:], o2, do, lse2, dt, dt[D:], dt[2 * D:], 3 * D, 3 * D, 3 * D, 3 * D, 3 * D, 3 * D, B, H, T, T, DH,
                        ops.MASK_KEYPAD, lens, 0.1, 7, keep_bits=bits)
            outs.append(dqkv)
    finally:
        assert C.neurst_set_option(b'mha_fused', 1) == 0
    assert torch.equal(outs[0], outs[1])
    a, b = outs[2].float(), outs[0].float()
    assert torch.isfinite(a).all() and ((a - b).abs().max() / b.abs().max()).item() < 2e-2
    assert ((a - b).norm() / b.norm()).item() < 1e-2
    pad = torch.arange(T, device=DEV)[None, :] >= lens[:, None]
    assert (outs[2][:, :, D:][pad] == 0).all()                # keys behind the length: exactly zero dK, dV


def test_ctc_gradient_frames_sum_to_zero_and_padding_is_zero():
    rng = np.random.default_rng(3)
    logits = torch.randn(B, T, V, device=DEV) * 2.0
    len_logits = torch.from_numpy(rng.integers(T // 2, T + 1, size=B).astype(np.int32)).to(DEV)
    U = 40
    labels = torch.from_numpy(rng.integers(4, V - 1, size=(B, U)).astype(np.int32)).to(DEV)
    len_labels = torch.from_numpy(rng.integers(U // 2, U + 1, size=B).astype(np.int32)).to(DEV)
    ref_argmax = logits.argmax(-1).int()
    loss, grad, argmax = ops.ctc_loss_fwd_bwd(logits, len_logits, labels, len_labels, ops.F32, 1.0)
    torch.cuda.synchronize()
    assert torch.isfinite(loss).all() and (loss > 0).all()
    pad = torch.arange(T, device=DEV)[None, :] >= len_logits[:, None]
    valid = ~pad
    assert torch.equal(argmax[valid], ref_argmax[valid])      # frames past the length report the blank
    frame_sum = grad.sum(-1)                                  # softmax - posterior: both sum to one over the vocabulary
    assert frame_sum.abs().max().item() < 5e-4
    assert grad[pad].abs().max().item() == 0.0
    assert (grad[valid].abs().sum(-1) > 0).all()


def test_layernorm_rows_are_standardised_and_backward_is_orthogonal():
    rows = B * T
    x = (torch.randn(rows, D, device=DEV) * 3 + 1).to(torch.bfloat16)
    y = torch.randn(rows, D, device=DEV).to(torch.bfloat16)
    gamma, beta = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    out, mean, rstd = ops.dropout_add_layernorm_fwd(x, y, gamma, beta, 0.0, 5)
    o = out.float()
    assert o.mean(-1).abs().max().item() < 2e-2 and (o.var(-1, unbiased=False) - 1).abs().max().item() < 3e-2
    # d/dz of LN is orthogonal to the constant vector and to x-hat: rows of dz sum to ~0 and are ~uncorrelated with out
    dout = torch.randn(rows, D, device=DEV).to(torch.bfloat16)
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    dz, _ = ops.dropout_add_layernorm_bwd(dout, y, gamma, mean, rstd, dg, db, 0.0, 5)
    dzf = dz.float()
    scale = dzf.abs().mean().item()
    assert dzf.mean(-1).abs().max().item() < 2e-2 * scale * 10
    assert (dzf * o).mean(-1).abs().max().item() < 5e-2 * scale * 10
    assert torch.allclose(db, dout.float().sum(0), rtol=1e-3, atol=1e-2)      # dbeta = column sums (atomics from 256 workgroups)
    assert torch.allclose(dg, (dout.float() * o).sum(0), rtol=2e-2, atol=2.0)


@pytest.mark.parametrize('ta,tb,M,N,K', [(0, 1, 16000, 2304, 768), (0, 0, 16000, 768, 3072), (1, 0, 3072, 768, 16000)])
def test_gemm_is_linear_in_its_operand(ta, tb, M, N, K):
    """operands on a 1/8 grid are exact in bf16 and their products/sums exact in fp32: C(A, B1 + B2) == C(A, B1) + C(A, B2)"""
    rng = np.random.default_rng(K)
    grid = lambda *s: torch.from_numpy(rng.integers(-8, 9, size=s).astype(np.float32) / 8).to(DEV).to(torch.bfloat16)
    A = grid(K, M) if ta else grid(M, K)
    B1, B2 = (grid(N, K), grid(N, K)) if tb else (grid(K, N), grid(K, N))
    B12 = (B1.float() + B2.float()).to(torch.bfloat16)
    assert torch.equal(B12.float(), B1.float() + B2.float())
    outs = []
    for Bm in (B1, B2, B12):
        C = torch.zeros(M, N, device=DEV, dtype=torch.float32)
        sk = ops.pick_split_k(M, N, K) if ta else 1
        ops.gemm(A, Bm, C, M, N, K, A.shape[1], Bm.shape[1], N, ta, tb, accumulate=sk > 1, split_k=sk)
        outs.append(C)
    assert torch.equal(outs[0] + outs[1], outs[2])            # every partial sum is a multiple of 1/64 below 2^24: exact


def _lut_model(num_blocks=None, dropout=None, dtype='bf16', V_=V):
    from neurst_amd.st.layers.arguments import build_args, load_config
    from neurst_amd.st.models.TransformerTPE import GlobalStep
    cfg = load_config(os.path.join(ge.ROOT, 'configs', 'lut.yaml'))
    for part in ('encoder', 'encoder2', 'decoder'):
        if num_blocks:
            cfg['model'][part]['num_blocks'] = num_blocks
        if dropout is not None:
            cfg['model'][part]['attention_dropout_rate'] = dropout
            cfg['model'][part]['residual_dropout_rate'] = dropout
    if dropout is not None:
        cfg['model']['kd']['attention_dropout_rate'] = dropout
    cfg['beam_size'] = 1
    args = build_args(cfg, dim_output=V_, compute_dtype=dtype)
    return args.Model(GlobalStep(1000), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args), args


def test_config2_forward_is_deterministic_and_gradients_reproduce():
    """BASELINE config 2 as bench.py runs it (dropout 0.1, counter-based masks): the same step twice gives bit-identical
    losses (no atomics on the forward path) and gradients that differ only by fp32 atomic-add ordering."""
    model, _ = _lut_model()
    batch = make_batch(B=B, Traw=1500, V=V, L=40, U=40, Lb=40, bert_dim=768, seed=11, ragged=True)
    batch = {k: torch.from_numpy(v).to(DEV) for k, v in batch.items()}
    a = model.forward_backward(batch).clone()
    ga = model.store.grads.clone()
    b = model.forward_backward(batch).clone()
    gb = model.store.grads
    assert torch.isfinite(a).all() and torch.equal(a, b)
    assert torch.isfinite(ga).all()
    den = ga.norm().item()
    assert den > 0 and (ga - gb).norm().item() < 1e-4 * den


def test_mean_of_tower_gradients_equals_full_batch_gradient():
    """gradientTools.py:7-37 / lstmModel.py:342-350: the batch is split into equal towers after padding and the tower
    gradients are averaged.  Two half-batch passes through the same kernels must reproduce the full-batch gradient
    (dropout off; LUT widths, 2 blocks each, bf16: agreement to bf16 rounding of the activations)."""
    model, _ = _lut_model(num_blocks=2, dropout=0.0, V_=1000)
    Bt = 8
    host = make_batch(B=Bt, Traw=600, V=1000, L=20, U=16, Lb=12, bert_dim=768, seed=5, ragged=True)
    dev = {k: torch.from_numpy(v).to(DEV) for k, v in host.items()}
    full_loss = model.forward_backward(dev).clone()
    full = model.store.grads.clone()
    acc = torch.zeros_like(full)
    losses = []
    for h in range(2):
        half = {k: v[h * Bt // 2:(h + 1) * Bt // 2].contiguous() for k, v in dev.items()}
        losses.append(model.forward_backward(half).clone())
        acc += model.store.grads
    acc *= 0.5
    assert torch.allclose((losses[0] + losses[1]) * 0.5, full_loss, rtol=2e-3, atol=1e-3), (losses, full_loss)
    rel = (acc - full).norm().item() / full.norm().item()
    cos = torch.dot(acc, full).item() / (acc.norm().item() * full.norm().item())
    assert rel < 2e-2 and cos > 0.9995, (rel, cos)


def _costt_model(num_blocks=None, dropout=None, dtype='bf16', V_=V, specaug=True):
    """configs/costt.yaml (= configs_template/costt.yaml dims): d = 512, 8 heads of 64, 6 + 6 + 6 blocks, shrink layer, SpecAugment"""
    from neurst_amd.st.layers.arguments import build_args, load_config
    from neurst_amd.st.models.TransformerTPE import GlobalStep
    cfg = load_config(os.path.join(ge.ROOT, 'configs', 'costt.yaml'))
    for part in ('encoder', 'encoder2', 'decoder'):
        if num_blocks:
            cfg['model'][part]['num_blocks'] = num_blocks
        if dropout is not None:
            cfg['model'][part]['attention_dropout_rate'] = dropout
            cfg['model'][part]['residual_dropout_rate'] = dropout
    cfg['model']['use_specAug'] = specaug
    cfg['beam_size'] = 1
    args = build_args(cfg, dim_output=V_, compute_dtype=dtype)
    assert args.model.shrink_layer and args.model.loss_type == 'ce_ctc' and args.model.encoder.num_cell_units == 512
    return args.Model(GlobalStep(1000), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args), args


def test_config3_costt_step_is_deterministic_at_full_width():
    """BASELINE config 3 as `bench.py --config configs/costt.yaml` runs it (B = 32 x 1500 raw frames, L = 80, dropout 0.1,
    SpecAugment, shrink layer): the same step twice gives bit-identical losses and gradients equal up to fp32 atomic order;
    every value finite; the shrunk length stays within the frame count."""
    model, _ = _costt_model()
    batch = make_batch(B=B, Traw=1500, V=V, L=80, U=40, seed=12, ragged=True)
    batch = {k: torch.from_numpy(v).to(DEV) for k, v in batch.items()}
    a = model.forward_backward(batch).clone()
    ga = model.store.grads.clone()
    len_a = model.last['len_encoded'].clone()
    b = model.forward_backward(batch).clone()
    gb = model.store.grads
    assert torch.isfinite(a).all() and torch.equal(a, b)
    assert torch.isfinite(ga).all()
    den = ga.norm().item()
    assert den > 0 and (ga - gb).norm().item() < 1e-4 * den
    assert torch.equal(len_a, model.last['len_encoded']) and int(len_a.max()) <= 500 and int(len_a.min()) >= 0


def test_config3_costt_mean_of_tower_gradients_equals_full_batch_gradient():
    """the reference's data-parallel rule (gradientTools.py:7-37) on the COSTT graph at its template width (2 blocks each,
    dropout and SpecAugment off, fp32 mode so that the arg-max decisions of the shrink layer are the same in both runs): the
    shrunk tensors of the half batches are padded to their own maxima, the gradient must not see the difference."""
    model, _ = _costt_model(num_blocks=2, dropout=0.0, dtype='fp32', V_=1000, specaug=False)
    Bt = 8
    host = make_batch(B=Bt, Traw=600, V=1000, L=30, U=16, seed=5, ragged=True)
    dev = {k: torch.from_numpy(v).to(DEV) for k, v in host.items()}
    full_loss = model.forward_backward(dev).clone()
    full = model.store.grads.clone()
    acc = torch.zeros_like(full)
    losses = []
    for h in range(2):
        half = {k: v[h * Bt // 2:(h + 1) * Bt // 2].contiguous() for k, v in dev.items()}
        losses.append(model.forward_backward(half).clone())
        acc += model.store.grads
    acc *= 0.5
    assert torch.allclose((losses[0] + losses[1]) * 0.5, full_loss, rtol=1e-5, atol=1e-6), (losses, full_loss)
    rel = (acc - full).norm().item() / full.norm().item()
    assert rel < 1e-4, rel


def test_shrink_gather_scatter_round_trip_at_config3_size():
    """blocks.py:8-38 at T = 500, d = 512, B = 32: scatter(gather(x)) returns x on the kept frames and zero elsewhere,
    gather(scatter(y)) returns y on the first new_len rows; kept frames are exactly the non-blank, non-repeated arg-maxes."""
    rng = np.random.default_rng(3)
    Bq, T, D = 32, 500, 512
    blank = 99
    am = rng.integers(90, 100, size=(Bq, T)).astype(np.int32)            # many blanks and repeats
    lens = rng.integers(250, T + 1, size=Bq).astype(np.int32)
    lens[0] = T
    am[1, :] = blank                                                       # an utterance that keeps nothing
    x = torch.from_numpy(rng.standard_normal((Bq, T, D)).astype(np.float32)).to(DEV).to(torch.bfloat16)
    index, new_len = ops.shrink_index(torch.from_numpy(am).to(DEV), torch.from_numpy(lens).to(DEV), blank)
    prev = np.concatenate([np.full((Bq, 1), blank, np.int32), am[:, :-1]], 1)
    keep = (am != blank) & (am != prev) & (np.arange(T)[None] < lens[:, None])
    assert np.array_equal(new_len.cpu().numpy(), keep.sum(1)) and int(new_len[1]) == 0
    Tout = max(int(new_len.max()), 1)
    y = ops.shrink_fwd(x, index, Tout)
    back = ops.shrink_bwd(y, index, T)
    kd = torch.from_numpy(keep).to(DEV)
    assert torch.equal(back[kd], x[kd]) and float(back[~kd].float().abs().max()) == 0.0
    y2 = ops.shrink_fwd(back, index, Tout)
    assert torch.equal(y2, y)
    valid = torch.arange(Tout, device=DEV)[None] < new_len[:, None]
    assert float(y[~valid].float().abs().max()) == 0.0


def test_beam_kernels_at_full_vocabulary_and_batch():
    """beam search bookkeeping at LUT scale (32 utterances x 4 hypotheses, V = 10000): top-k agrees with torch.topk of
    the log-softmax, a pruning step keeps the best candidates in order, and parents / gathers stay inside the utterance"""
    beam, R = 4, B * 4
    g = torch.Generator(device='cpu').manual_seed(5)
    logits = (torch.randn(R, V, generator=g) * 3).to(DEV)
    sc, ids = ops.beam_topk(logits, beam)
    ref = torch.log_softmax(logits.double(), -1)
    rs, ri = ref.topk(beam, dim=-1)
    assert torch.equal(ids.long(), ri)                        # random fp32 logits: no ties
    assert (sc.double() - rs).abs().max().item() < 1e-5
    scores = torch.tensor(([0.0] + [-1e10] * (beam - 1)) * B, device=DEV)
    len_dec = torch.ones(R, dtype=torch.int32, device=DEV)
    fin = torch.zeros(R, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(3, dtype=torch.int32, device=DEV)
    for step in range(3):                                     # three steps: from one live hypothesis to a full beam
        n_sc, n_len, n_fin, tok, par = ops.beam_prune(sc, ids, scores, len_dec, fin, cnt[step:step + 1], beam, 3, 0.2,
                                                      prev_unfinished=cnt[step - 1:step] if step else None)
        s2 = n_sc.view(B, beam)
        assert (s2[:, :-1] >= s2[:, 1:]).all()                # survivors come out best first
        assert torch.equal(par // beam, torch.arange(R, device=DEV, dtype=torch.int32) // beam)      # parents stay in their utterance
        cand = ((scores[:, None] + sc) / (((5.0 + (len_dec[:, None] + 1).double()) / 6.0) ** 0.2).float()).view(B, beam * beam)
        assert torch.allclose(s2[:, 0], cand.max(-1).values, rtol=1e-6)
        if step == 0:
            assert (par % beam == 0).all()                    # only hypothesis 0 was alive
        scores, len_dec, fin = n_sc, n_len, n_fin
        logits = ops.gather_rows(logits, par) + 0.1 * step    # as the caches are re-ordered in the real loop
        sc, ids = ops.beam_topk(logits, beam)
    assert int(cnt[2].item()) == int((fin == 0).sum().item()) and (len_dec >= 1).all() and (len_dec <= 4).all()


def test_batch_past_32bit_indexing_gives_the_gradient_of_its_repeated_part():
    """BASELINE config 5 (long utterances, batch sized for the 288 GB of HBM): 32 utterances of 3000 raw frames repeated
    8 times - B = 256, T = 1000, 2.56e9 CTC logits (past 32-bit element indices), ~90 GB of activations - must give the
    losses and the gradient of the 32 (every loss is a mean over utterances; dropout off)."""
    model, _ = _lut_model(dropout=0.0)
    host = make_batch(B=32, Traw=3000, V=V, L=40, U=40, Lb=40, bert_dim=768, seed=11, ragged=True)
    small = {k: torch.from_numpy(v).to(DEV) for k, v in host.items()}
    a = model.forward_backward(small).clone()
    ga = model.store.grads.clone()
    big = {k: v.repeat((8,) + (1,) * (v.dim() - 1)).contiguous() for k, v in small.items()}
    assert big['feature'].shape[0] * 1000 * V > 2 ** 31
    b = model.forward_backward(big).clone()
    gb = model.store.grads
    assert torch.isfinite(b).all() and torch.allclose(a, b, rtol=2e-3), (a, b)
    rel = (gb - ga).norm().item() / ga.norm().item()
    assert rel < 1e-2, rel
    del big, small, model
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------------------------
# BASELINE config 5: long utterances (configs_template/lut.yaml:101-107 - the last bucket boundary is 2050 frames after the
# 3 x down-sampling; Traw = 6000 -> T = 2000).  The oracle would take minutes here: properties instead.

def test_config5_long_utterance_step_is_finite_and_ctc_frames_sum_to_zero():
    """LUT at the template's width and depth (4 + 4 + 4 blocks, bf16, dropout 0.1) on B = 8 utterances of 6000 raw frames
    (T = 2000 encoder frames, ragged): every loss and every gradient finite, padded CTC frames contribute nothing; and the
    CTC kernels alone at [4, 2000, 10000]: a gradient frame sums to zero, padding is exactly zero."""
    model, _ = _lut_model()
    Bl, Traw = 8, 6000
    host = make_batch(B=Bl, Traw=Traw, V=V, L=40, U=40, Lb=40, bert_dim=768, seed=21, ragged=True)
    batch = {k: torch.from_numpy(v).to(DEV) for k, v in host.items()}
    out = model.forward_backward(batch)
    torch.cuda.synchronize()
    assert tuple(model.last['shape_batch'])[:2] == (Bl, 2000)
    assert torch.isfinite(out).all() and float(out[0]) > 0
    g = model.store.grads
    assert torch.isfinite(g).all() and g.abs().max().item() > 0
    loss2 = model.train_step(batch)[0]
    assert np.isfinite(loss2)
    del model, g
    torch.cuda.empty_cache()
    rng = np.random.default_rng(5)
    Tl = 2000
    logits = torch.randn(4, Tl, V, device=DEV) * 2.0
    len_logits = torch.from_numpy(rng.integers(Tl // 2, Tl + 1, size=4).astype(np.int32)).to(DEV)
    labels = torch.from_numpy(rng.integers(4, V - 1, size=(4, 40)).astype(np.int32)).to(DEV)
    len_labels = torch.from_numpy(rng.integers(20, 41, size=4).astype(np.int32)).to(DEV)
    loss, grad, _ = ops.ctc_loss_fwd_bwd(logits, len_logits, labels, len_labels, ops.F32, 1.0)
    pad = torch.arange(Tl, device=DEV)[None, :] >= len_logits[:, None]
    assert torch.isfinite(loss).all() and (loss > 0).all()
    assert grad.sum(-1).abs().max().item() < 5e-4 and grad[pad].abs().max().item() == 0.0


def test_attention_beyond_the_fused_kernels_query_limit_falls_back_to_the_tiled_pair():
    """The one-kernel backward keeps the statistics of every query in LDS (MAX_TQ = 2048); longer inputs must take the tiled
    dQ + dK/dV pair without being asked.  T = 2200 with every key at and behind 2000 masked and dO = 0 for the queries
    behind 2000 is the SAME problem as T = 2000 on the truncated tensors, which the fused kernel takes: forward rows still
    sum to one, and dQ / dK / dV of the first 2000 positions agree to bf16 rounding between the two paths; the keys
    behind the length get exactly zero gradients."""
    Bs, Tl, Ts = 2, 2200, 2000
    g = torch.Generator(device='cpu').manual_seed(9)
    qkv = (torch.randn(Bs, Tl, 3 * D, generator=g) * 0.5).to(DEV).to(torch.bfloat16)
    lens = torch.full((Bs,), Ts, dtype=torch.int32, device=DEV)
    t = qkv.view(-1)
    ones = qkv.clone()
    ones[:, :, 2 * D:] = 1.0
    to = ones.view(-1)
    o1, lse1 = ops.mha_fwd(to, to[D:], to[2 * D:], 3 * D, 3 * D, 3 * D, Bs, H, Tl, Tl, DH, ops.MASK_KEYPAD, lens, 0.0, 3)
    assert torch.isfinite(lse1).all() and (o1.float() - 1.0).abs().max().item() < 1e-2
    o, lse = ops.mha_fwd(t, t[D:], t[2 * D:], 3 * D, 3 * D, 3 * D, Bs, H, Tl, Tl, DH, ops.MASK_KEYPAD, lens, 0.0, 3)
    do = torch.randn(Bs, Tl, D, generator=g).to(DEV).to(torch.bfloat16)
    do[:, Ts:] = 0
    dqkv = torch.zeros_like(qkv)
    dt = dqkv.view(-1)
    ops.mha_bwd(t, t[D:], t[2 * D:], o, do, lse, dt, dt[D:], dt[2 * D:], 3 * D, 3 * D, 3 * D, 3 * D, 3 * D, 3 * D, Bs, H, Tl, Tl, DH,
                ops.MASK_KEYPAD, lens, 0.0, 3)
    # the truncated problem (fused kernels: Tq = 2000 <= MAX_TQ)
    qs = qkv[:, :Ts].contiguous()
    ts_ = qs.view(-1)
    o_s, lse_s = ops.mha_fwd(ts_, ts_[D:], ts_[2 * D:], 3 * D, 3 * D, 3 * D, Bs, H, Ts, Ts, DH, ops.MASK_KEYPAD, lens, 0.0, 3)
    dqs = torch.zeros_like(qs)
    dts = dqs.view(-1)
    ops.mha_bwd(ts_, ts_[D:], ts_[2 * D:], o_s, do[:, :Ts].contiguous(), lse_s, dts, dts[D:], dts[2 * D:], 3 * D, 3 * D, 3 * D, 3 * D, 3 * D, 3 * D,
                Bs, H, Ts, Ts, DH, ops.MASK_KEYPAD, lens, 0.0, 3)
    torch.cuda.synchronize()
    assert ((o[:, :Ts].float() - o_s.float()).abs().max() / o_s.float().abs().max()).item() < 2e-2
    a, b = dqkv[:, :Ts].float(), dqs.float()
    assert torch.isfinite(a).all() and torch.isfinite(b).all() and b.abs().max().item() > 0
    assert ((a - b).norm() / b.norm()).item() < 1e-2 and ((a - b).abs().max() / b.abs().max()).item() < 3e-2
    assert (dqkv[:, Ts:, D:] == 0).all()                      # masked keys: exactly zero dK, dV
