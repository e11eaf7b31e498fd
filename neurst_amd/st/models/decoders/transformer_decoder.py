"""Transformer decoder, st/models/decoders/transformer_decoder.py: teacher-forced `decode` /
`decoder_impl` (:36-50, :184-246), the greedy `decoder_with_caching` (:52-182) and `beam_decode_rerank` (:248-431)."""
import numpy as np
import torch

from .decoder import Decoder
from .. import graph as G
from ..sites import SITE_DEC_IN, SITE_DEC_BLK, blk_site
from ..tools.utils import dense, residual, multihead_attention, ff_hidden, tf_layers_dense
from ...layers import common_attention_v1
from .... import ops


class Transformer_Decoder(Decoder):
    def __init__(self, args, is_train, global_step, embed_table=None, name=None):
        super().__init__(args, is_train, global_step, embed_table, name)
        d = args.model.decoder
        self.num_blocks = d.num_blocks
        self.num_cell_units = d.num_cell_units
        self.attention_dropout_rate = d.attention_dropout_rate if is_train else 0.0
        self.residual_dropout_rate = d.residual_dropout_rate if is_train else 0.0
        self.num_heads = d.num_heads
        self.size_embedding = d.size_embedding

    def decode(self, encoded, len_encoded, decoder_input):
        """MLE training: logits [B, L, V] (fp32)."""
        decoder_output = self.decoder_impl(decoder_input, encoded, len_encoded)
        logits = tf_layers_dense(decoder_output, self.args.dim_output, use_bias=False, name='decoder_fc', out_dtype=ops.F32)
        return logits, None, len_encoded

    def decoder_impl(self, decoder_input, encoder_output, len_encoded):
        encoder_attention_bias = common_attention_v1.attention_bias_ignore_padding(len_encoded)
        # embedding, `x += x + timing signal` (sic, :195) and dropout in one kernel
        decoder_output = self.embedding(decoder_input, dropout_rate=self.residual_dropout_rate, site=SITE_DEC_IN)
        # causal bias only - the reference adds no target-padding bias (:201, :211)
        self_attention_bias = common_attention_v1.attention_bias_lower_triangle(decoder_input.shape[1])
        d = self.num_cell_units
        for i in range(self.num_blocks):
            with G.variable_scope('block_{}'.format(i)):
                decoder_output = residual(decoder_output,
                                          multihead_attention(query_antecedent=decoder_output, memory_antecedent=None,
                                                              bias=self_attention_bias, total_key_depth=d, total_value_depth=d,
                                                              num_heads=self.num_heads, dropout_rate=self.attention_dropout_rate,
                                                              output_depth=d, name='decoder_self_attention', summaries=True,
                                                              site=blk_site(SITE_DEC_BLK, i, 0)),
                                          dropout_rate=self.residual_dropout_rate, site=blk_site(SITE_DEC_BLK, i, 1))
                decoder_output = residual(decoder_output,
                                          multihead_attention(query_antecedent=decoder_output, memory_antecedent=encoder_output,
                                                              bias=encoder_attention_bias, total_key_depth=d, total_value_depth=d,
                                                              output_depth=d, num_heads=self.num_heads,
                                                              dropout_rate=self.attention_dropout_rate,
                                                              name='decoder_vanilla_attention', summaries=True,
                                                              site=blk_site(SITE_DEC_BLK, i, 2)),
                                          dropout_rate=self.residual_dropout_rate, site=blk_site(SITE_DEC_BLK, i, 3))
                decoder_output = residual(decoder_output,
                                          ff_hidden(decoder_output, hidden_size=4 * d, output_size=d, activation='relu'),
                                          dropout_rate=self.residual_dropout_rate, site=blk_site(SITE_DEC_BLK, i, 4))
        return decoder_output

    def decoder_with_caching(self, encoded, len_encoded):
        """Greedy search (:52-121).  The reference re-embeds the whole prefix every step and caches block
        OUTPUTS; here each block caches its self-attention K/V projections (and the encoder K/V are
        projected once), which computes the same values for the last position.  Returns
        (None, preds [B, steps] int32 zeroed past len_decoded, len_decoded)."""
        g = G.current()
        assert g.tape is None
        B, T = encoded.shape[0], encoded.shape[1]
        dev = encoded.device
        max_steps = int(min(T + 50, self.args.max_len))
        preds = torch.zeros((B, max_steps), dtype=torch.int32, device=dev)
        cur = torch.full((B, 1), int(self.start_token), dtype=torch.int32, device=dev)   # token fed at this step
        finished = torch.zeros(B, dtype=torch.int32, device=dev)
        len_decoded = torch.ones(B, dtype=torch.int32, device=dev)
        unfinished_hist = torch.zeros(max_steps, dtype=torch.int32, device=dev)   # rows still decoding after each step
        CHECK_EVERY = 8      # host looks at the stop condition every 8 steps (one read back), not every step
        run = _CachedDecoderStep(self, encoded, len_encoded, beam=1, max_steps=max_steps)
        step = 0
        while step < max_steps:
            cur_logit = run.step(cur, step)
            ops.greedy_step(cur_logit, preds, finished, len_decoded, unfinished_hist[step:step + 1], step, self.end_token)
            cur.copy_(preds[:, step:step + 1])
            step += 1
            # tf.reduce_any(not finished): the loop ends after the first step that leaves no row unfinished.  Steps run
            # past that point (at most CHECK_EVERY - 1) only touch finished rows and are trimmed below.
            if step % CHECK_EVERY == 0 or step == max_steps:
                done = np.nonzero(unfinished_hist[:step].cpu().numpy() == 0)[0]
                if done.size:
                    step = int(done[0]) + 1
                    break
        fin = finished.cpu().numpy()
        ld = len_decoded.cpu().numpy() - (1 - fin)        # "for decoded length cut by encoded length" (:115)
        out = preds[:, :step].cpu().numpy()
        out[np.arange(step)[None, :] >= ld[:, None]] = 0   # preds * sequence_mask(len_decoded) (:118-119)
        return None, out, ld

    def beam_decode_rerank(self, encoded, len_encoded):
        """Beam search with length penalty (:248-431; lambda_lm plays no part there: z_lm is zeros, :309).  Rows are
        (utterance b, hypothesis j) = b * beam + j.  Per step: the cached decoder on the last token of every hypothesis
        (self-attention over the hypothesis' own K/V cache; cross-attention as `beam` queries per utterance against the
        utterance's encoder K/V, projected once), then three kernels - top-k of the log-softmax, pruning of the beam^2
        candidates per utterance, re-ordering of predictions and caches by parent - with no read-back; the stop
        condition is read every 8 steps as in the greedy loop.
        Returns ([preds_sorted [B, beam, steps], scores_sorted [B, beam], scores_sorted], best preds [B, steps] int32
        (as the reference: NOT masked past <eos>), len_decoded [B * beam] in unsorted beam order (:431))."""
        g = G.current()
        assert g.tape is None
        K = int(self.args.beam_size)
        lpw = float(self.args.length_penalty_weight or 0.0)
        B, T = encoded.shape[0], encoded.shape[1]
        R = B * K
        d = self.num_cell_units
        dev = encoded.device
        max_steps = int(min(T + 50, self.args.max_len))
        INF = 1e10
        preds = torch.zeros((R, max_steps), dtype=torch.int32, device=dev)
        preds_next = torch.zeros_like(preds)
        cur = torch.full((R, 1), int(self.start_token), dtype=torch.int32, device=dev)
        finished = torch.zeros(R, dtype=torch.int32, device=dev)
        len_decoded = torch.ones(R, dtype=torch.int32, device=dev)
        scores = torch.tensor(([0.0] + [-INF] * (K - 1)) * B, dtype=torch.float32, device=dev)
        unfinished_hist = torch.zeros(max_steps, dtype=torch.int32, device=dev)
        CHECK_EVERY = 8
        run = _CachedDecoderStep(self, encoded, len_encoded, beam=K, max_steps=max_steps)
        kv_next = [torch.empty_like(c) for c in run.kv_cache]
        step = 0
        while step < max_steps:
            cur_logit = run.step(cur, step)
            next_scores, next_ids = ops.beam_topk(cur_logit, K)
            scores, len_decoded, finished, tokens, parents = ops.beam_prune(next_scores, next_ids, scores, len_decoded, finished,
                                                                            unfinished_hist[step:step + 1], K, int(self.end_token), lpw,
                                                                            unfinished_hist[step - 1:step] if step else None)
            ops.gather_rows(preds, parents, out=preds_next)
            preds, preds_next = preds_next, preds
            preds[:, step].copy_(tokens)
            row_bytes = (step + 1) * 2 * d * run.kv_cache[0].element_size()      # only the step + 1 positions written so far
            for i in range(self.num_blocks):
                ops.gather_rows(run.kv_cache[i], parents, out=kv_next[i], row_bytes=row_bytes)
            kv_next = run.swap_caches(kv_next)
            cur.copy_(tokens.view(R, 1))
            step += 1
            if step % CHECK_EVERY == 0 or step == max_steps:
                done = np.nonzero(unfinished_hist[:step].cpu().numpy() == 0)[0]
                if done.size:      # steps enqueued past the stopping step were copied through unchanged by beam_prune
                    step = int(done[0]) + 1
                    break
        return self._beam_result(preds, scores, len_decoded, finished, step, B, K)

    @staticmethod
    def _beam_result(preds, scores, len_decoded, finished, step, B, K):
        sc = scores.cpu().numpy().reshape(B, K)
        order = np.argsort(-sc, axis=1, kind='stable')                   # tf.nn.top_k(sorted=True): ties by lower index
        rows = (np.arange(B)[:, None] * K + order).reshape(-1)
        p = preds[:, :step].cpu().numpy()[rows].reshape(B, K, step)
        s_sorted = np.take_along_axis(sc, order, 1)
        ld = len_decoded.cpu().numpy() - (1 - finished.cpu().numpy())
        return [p, s_sorted, s_sorted], p[:, 0, :], ld


class _CachedDecoderStep:
    """decoder_with_caching_impl + decoder_fc on the last position (:123-182, :65-75) for R = B * beam rows, for the decoding
    loops: the variables are looked up and every buffer is allocated ONCE, a step is ~55 launches through the C ABI with
    no allocation and no scope bookkeeping (the layer functions of tools/utils.py cost ~2x the host time per launch,
    and a decoding step is bound by exactly that).  Self-attention K/V projections go straight into the per-block cache
    [R, max_steps, 2d] (row `step`); the encoder-side K/V of every block are projected once per utterance, and the `beam`
    hypotheses of an utterance attend to them as `beam` queries of one batch entry."""

    def __init__(self, dec, encoded, len_encoded, beam, max_steps):
        g = G.current()
        self.g, self.dec, self.beam, self.max_steps = g, dec, beam, max_steps
        self.len_encoded = len_encoded
        B, T = encoded.shape[0], encoded.shape[1]
        d, H = dec.num_cell_units, dec.num_heads
        self.B, self.T, self.d, self.H, self.dh = B, T, d, H, d // H
        R = self.R = B * beam
        dev, tdt = encoded.device, ops.torch_dtype(g.dtype)
        V = self.V = dec.dim_output

        def lin(name, n_out, n_in):
            with G.variable_scope(name):
                return g.get_variable('kernel', [n_out, n_in]), g.get_variable('bias', [n_out])

        def ln(name):
            with G.variable_scope(name):
                return g.get_variable('layer_norm_scale', [d]).master, g.get_variable('layer_norm_bias', [d]).master
        self.blocks, self.mem_kv = [], []
        for i in range(dec.num_blocks):
            with G.variable_scope('block_{}'.format(i)):
                with G.variable_scope('decoder_self_attention'):
                    qkv_w, qkv_b = lin('qkv_transform', 3 * d, d)
                    so_w, so_b = lin('output_transform', d, d)
                with G.variable_scope('decoder_vanilla_attention'):
                    q_w, q_b = lin('q_transform', d, d)
                    co_w, co_b = lin('output_transform', d, d)
                    self.mem_kv.append(dense(encoded, 2 * d, name='kv_transform').view(-1))
                with G.variable_scope('ff_hidden'):
                    f1_w, f1_b = lin('dense', 4 * d, d)
                    f2_w, f2_b = lin('dense_1', d, 4 * d)
                blk = dict(q_w=qkv_w.w[:d], q_b=qkv_b.master[:d], kv_w=qkv_w.w[d:], kv_b=qkv_b.master[d:], so_w=so_w.w, so_b=so_b.master,
                           cq_w=q_w.w, cq_b=q_b.master, co_w=co_w.w, co_b=co_b.master, f1_w=f1_w.w, f1_b=f1_b.master, f2_w=f2_w.w,
                           f2_b=f2_b.master, ln=[ln('layer_norm'), ln('layer_norm_1'), ln('layer_norm_2')])
                self.blocks.append(blk)
            g.name_counts = {}
        with G.variable_scope('decoder_fc'):
            self.fc_w = g.get_variable('kernel', [d, V]).w
        new = lambda *shape, dt=tdt: torch.empty(shape, dtype=dt, device=dev)
        self.kv_cache = [new(R, max_steps, 2 * d) for _ in range(dec.num_blocks)]
        self.x = [new(R, 1, d), new(R, 1, d)]
        self.q, self.a, self.o, self.h = new(R, d), new(R, d), new(R, d), new(R, 4 * d)
        self.mean, self.rstd = new(R, dt=torch.float32), new(R, dt=torch.float32)
        self.lse = new(R * H, dt=torch.float32)
        self.logits = new(R, V, dt=torch.float32)

    def swap_caches(self, other):
        """beam search re-orders the caches into `other`; returns the old buffers for the next re-ordering"""
        old, self.kv_cache = self.kv_cache, other
        return old

    def step(self, cur, step):
        """cur [R, 1] int32: the token fed at this step -> fp32 logits [R, V] (a buffer that the next step overwrites)"""
        R, d, H, dh, T, K, ms = self.R, self.d, self.H, self.dh, self.T, self.beam, self.max_steps
        gemm, lnf = ops.gemm, ops.dropout_add_layernorm_fwd
        x, y = self.x
        q, a, o, h = self.q, self.a, self.o, self.h
        stats = (self.mean, self.rstd)
        self.dec.embedding(cur, dropout_rate=0.0, pos0=step, signal_len=ms, out=x)
        for blk, cache, mk in zip(self.blocks, self.kv_cache, self.mem_kv):
            kc = cache.view(-1)
            gemm(x, blk['q_w'], q, R, d, d, d, d, d, False, True, bias=blk['q_b'])
            gemm(x, blk['kv_w'], kc[step * 2 * d:], R, 2 * d, d, d, d, ms * 2 * d, False, True, bias=blk['kv_b'])
            ops.mha_fwd_strided(q, kc, kc[d:], a, self.lse, d, 2 * d, 2 * d, d, d, ms * 2 * d, ms * 2 * d, d, R, H, 1, step + 1, dh,
                                ops.MASK_NONE, None)
            gemm(a, blk['so_w'], o, R, d, d, d, d, d, False, True, bias=blk['so_b'])
            lnf(x, o, blk['ln'][0][0], blk['ln'][0][1], 0.0, 0, None, 1, bufs=(y,) + stats)
            x, y = y, x
            gemm(x, blk['cq_w'], q, R, d, d, d, d, d, False, True, bias=blk['cq_b'])
            # B "batches" of K queries each: the hypotheses of an utterance share its keys and values
            ops.mha_fwd_strided(q, mk, mk[d:], a, self.lse, d, 2 * d, 2 * d, d, K * d, T * 2 * d, T * 2 * d, K * d, self.B, H, K, T, dh,
                                ops.MASK_KEYPAD, self.len_encoded)
            gemm(a, blk['co_w'], o, R, d, d, d, d, d, False, True, bias=blk['co_b'])
            lnf(x, o, blk['ln'][1][0], blk['ln'][1][1], 0.0, 0, None, 1, bufs=(y,) + stats)
            x, y = y, x
            gemm(x, blk['f1_w'], h, R, 4 * d, d, d, d, 4 * d, False, True, bias=blk['f1_b'], relu=True)
            gemm(h, blk['f2_w'], o, R, d, 4 * d, 4 * d, 4 * d, d, False, True, bias=blk['f2_b'])
            lnf(x, o, blk['ln'][2][0], blk['ln'][2][1], 0.0, 0, None, 1, bufs=(y,) + stats)
            x, y = y, x
        self.x = [x, y]
        gemm(x, self.fc_w, self.logits, R, self.V, d, d, self.V, self.V, False, False)
        return self.logits
