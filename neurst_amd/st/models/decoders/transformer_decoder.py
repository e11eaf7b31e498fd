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
        d, H = self.num_cell_units, self.num_heads
        dh = d // H
        dev = encoded.device
        max_steps = int(min(T + 50, self.args.max_len))
        tdt = ops.torch_dtype(g.dtype)
        preds = torch.zeros((B, max_steps), dtype=torch.int32, device=dev)
        cur = torch.full((B, 1), int(self.start_token), dtype=torch.int32, device=dev)   # token fed at this step
        finished = torch.zeros(B, dtype=torch.int32, device=dev)
        len_decoded = torch.ones(B, dtype=torch.int32, device=dev)
        unfinished_hist = torch.zeros(max_steps, dtype=torch.int32, device=dev)   # rows still decoding after each step
        CHECK_EVERY = 8      # host looks at the stop condition every 8 steps (one read back), not every step
        kv_cache = [torch.empty((B, max_steps, 2 * d), dtype=tdt, device=dev) for _ in range(self.num_blocks)]
        mem_kv = []
        for i in range(self.num_blocks):   # encoder-side K/V of every block, once
            with G.variable_scope('block_{}'.format(i)), G.variable_scope('decoder_vanilla_attention'):
                mem_kv.append(dense(encoded, 2 * d, name='kv_transform'))
        lse_self = torch.empty((B, H, 1), dtype=torch.float32, device=dev)
        step = 0
        while step < max_steps:
            x = self.embedding(cur, dropout_rate=0.0, pos0=step, signal_len=max_steps)     # [B,1,d]
            for i in range(self.num_blocks):
                with G.variable_scope('block_{}'.format(i)):
                    with G.variable_scope('decoder_self_attention'):
                        qkv = dense(x, 3 * d, name='qkv_transform')                                # [B,1,3d]
                        kv_cache[i][:, step, :].copy_(qkv[:, 0, d:])                                # buffer plumbing (strided D2D copy)
                        a = torch.empty((B, 1, d), dtype=tdt, device=dev)
                        kc = kv_cache[i].view(-1)
                        ops.mha_fwd_strided(qkv.view(-1), kc, kc[d:], a, lse_self, 3 * d, 2 * d, 2 * d, d,
                                            3 * d, max_steps * 2 * d, max_steps * 2 * d, d, B, H, 1, step + 1, dh, ops.MASK_NONE, None)
                        o = dense(a, d, name='output_transform')
                    x = residual(x, o, 0.0)
                    with G.variable_scope('decoder_vanilla_attention'):
                        q = dense(x, d, name='q_transform')
                        a = torch.empty((B, 1, d), dtype=tdt, device=dev)
                        mk = mem_kv[i].view(-1)
                        ops.mha_fwd_strided(q.view(-1), mk, mk[d:], a, lse_self, d, 2 * d, 2 * d, d, d, T * 2 * d, T * 2 * d, d,
                                            B, H, 1, T, dh, ops.MASK_KEYPAD, len_encoded)
                        o = dense(a, d, name='output_transform')
                    x = residual(x, o, 0.0)
                    x = residual(x, ff_hidden(x, hidden_size=4 * d, output_size=d, activation='relu'), 0.0)
                    g.name_counts = {}   # the same layer_norm / ff_hidden names are reused every step
            cur_logit = tf_layers_dense(x.view(B, d), self.dim_output, use_bias=False, name='decoder_fc', out_dtype=ops.F32)
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
        d, H = self.num_cell_units, self.num_heads
        dh = d // H
        dev = encoded.device
        max_steps = int(min(T + 50, self.args.max_len))
        tdt = ops.torch_dtype(g.dtype)
        INF = 1e10
        preds = torch.zeros((R, max_steps), dtype=torch.int32, device=dev)
        preds_next = torch.zeros_like(preds)
        cur = torch.full((R, 1), int(self.start_token), dtype=torch.int32, device=dev)
        finished = torch.zeros(R, dtype=torch.int32, device=dev)
        len_decoded = torch.ones(R, dtype=torch.int32, device=dev)
        scores = torch.tensor(([0.0] + [-INF] * (K - 1)) * B, dtype=torch.float32, device=dev)
        unfinished_hist = torch.zeros(max_steps, dtype=torch.int32, device=dev)
        CHECK_EVERY = 8
        kv_cache = [torch.empty((R, max_steps, 2 * d), dtype=tdt, device=dev) for _ in range(self.num_blocks)]
        kv_next = [torch.empty_like(c) for c in kv_cache]
        mem_kv = []
        for i in range(self.num_blocks):   # encoder-side K/V of every block, once per utterance (shared by its hypotheses)
            with G.variable_scope('block_{}'.format(i)), G.variable_scope('decoder_vanilla_attention'):
                mem_kv.append(dense(encoded, 2 * d, name='kv_transform'))
        lse = torch.empty((R, H, 1), dtype=torch.float32, device=dev)
        step = 0
        while step < max_steps:
            x = self.embedding(cur, dropout_rate=0.0, pos0=step, signal_len=max_steps)     # [R,1,d]
            for i in range(self.num_blocks):
                with G.variable_scope('block_{}'.format(i)):
                    with G.variable_scope('decoder_self_attention'):
                        qkv = dense(x, 3 * d, name='qkv_transform')
                        kv_cache[i][:, step, :].copy_(qkv[:, 0, d:])
                        a = torch.empty((R, 1, d), dtype=tdt, device=dev)
                        kc = kv_cache[i].view(-1)
                        ops.mha_fwd_strided(qkv.view(-1), kc, kc[d:], a, lse, 3 * d, 2 * d, 2 * d, d,
                                            3 * d, max_steps * 2 * d, max_steps * 2 * d, d, R, H, 1, step + 1, dh, ops.MASK_NONE, None)
                        o = dense(a, d, name='output_transform')
                    x = residual(x, o, 0.0)
                    with G.variable_scope('decoder_vanilla_attention'):
                        q = dense(x, d, name='q_transform')
                        a = torch.empty((R, 1, d), dtype=tdt, device=dev)
                        mk = mem_kv[i].view(-1)
                        # B "batches" of K queries each: the hypotheses of an utterance share its keys and values
                        ops.mha_fwd_strided(q.view(-1), mk, mk[d:], a, lse, d, 2 * d, 2 * d, d, K * d, T * 2 * d, T * 2 * d, K * d,
                                            B, H, K, T, dh, ops.MASK_KEYPAD, len_encoded)
                        o = dense(a, d, name='output_transform')
                    x = residual(x, o, 0.0)
                    x = residual(x, ff_hidden(x, hidden_size=4 * d, output_size=d, activation='relu'), 0.0)
                    g.name_counts = {}
            cur_logit = tf_layers_dense(x.view(R, d), self.dim_output, use_bias=False, name='decoder_fc', out_dtype=ops.F32)
            next_scores, next_ids = ops.beam_topk(cur_logit, K)
            scores, len_decoded, finished, tokens, parents = ops.beam_prune(next_scores, next_ids, scores, len_decoded, finished,
                                                                            unfinished_hist[step:step + 1], K, int(self.end_token), lpw,
                                                                            unfinished_hist[step - 1:step] if step else None)
            ops.gather_rows(preds, parents, out=preds_next)
            preds, preds_next = preds_next, preds
            preds[:, step].copy_(tokens)
            for i in range(self.num_blocks):      # only the step + 1 positions written so far
                ops.gather_rows(kv_cache[i], parents, out=kv_next[i], row_bytes=(step + 1) * 2 * d * kv_cache[i].element_size())
            kv_cache, kv_next = kv_next, kv_cache
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
