"""TransformerTPE ("doubly supervised encoder" ST model used by LUT and COSTT),
st/models/TransformerTPE.py: ctc_loss (:74-104), ctc_decode (:106-125), build_graph (:127-193),
build_single_graph (:195-484), build_infer_graph (:486-513).

There is no TF session here: `train_step(batch)` runs what `sess.run(model.list_run + losses)` ran
(forward, backward, gradient averaging, Adam) as HIP kernel launches, `infer(features, len_feas)` runs
what `sess.run(model_infer.list_run, feed_dict)` ran.  `list_run` keeps the reference ordering as callables.
"""
import gc
import os
import weakref
from collections import namedtuple

import numpy as np
import torch

from . import graph as G
from .kd_layer import compute_kd_loss, extract_feat
from .seq2seqModel import Seq2SeqModel
from .tools.blocks import ShrinkSpeculation, shrink_layer
from ..tools.tftools.gradientTools import GradBucketer, average_gradients, handle_gradients
from ..tools.tftools.tfAudioTools import apply_specaugmant
from ..tools.tftools import tfRecord
from ... import ops


class GlobalStep:
    """Stand-in for tf.train.get_or_create_global_step()."""

    def __init__(self, value=0):
        self.value = int(value)


def variable_specs(args, name='transformer'):
    """(tf variable name, shape, init kind) in creation order - the scopes st/models/*.py open
    (SURVEY.md 8a parameter inventory)."""
    m = args.model
    V = args.dim_output
    specs = [('{}/embedding'.format(name), (V, m.decoder.size_embedding), 'glorot')]

    def attn_self(p, d):
        return [(p + '/qkv_transform/kernel', (3 * d, d), 'glorot'), (p + '/qkv_transform/bias', (3 * d,), 'zeros'),
                (p + '/output_transform/kernel', (d, d), 'glorot'), (p + '/output_transform/bias', (d,), 'zeros')]

    def attn_cross(p, d, dq, dm):
        return [(p + '/q_transform/kernel', (d, dq), 'glorot'), (p + '/q_transform/bias', (d,), 'zeros'),
                (p + '/kv_transform/kernel', (2 * d, dm), 'glorot'), (p + '/kv_transform/bias', (2 * d,), 'zeros'),
                (p + '/output_transform/kernel', (d, d), 'glorot'), (p + '/output_transform/bias', (d,), 'zeros')]

    def ln(p, idx, d):
        s = 'layer_norm' if idx == 0 else 'layer_norm_{}'.format(idx)
        return [('{}/{}/layer_norm_scale'.format(p, s), (d,), 'ones'), ('{}/{}/layer_norm_bias'.format(p, s), (d,), 'zeros')]

    def ffn(p, d):
        return [(p + '/ff_hidden/dense/kernel', (4 * d, d), 'glorot'), (p + '/ff_hidden/dense/bias', (4 * d,), 'zeros'),
                (p + '/ff_hidden/dense_1/kernel', (d, 4 * d), 'glorot'), (p + '/ff_hidden/dense_1/bias', (d,), 'zeros')]

    din = args.data.dim_input
    de = m.encoder.num_cell_units
    if m.use_cnn_resample:
        cin = args.data.feature_channel
        f = din // cin
        kernel = tuple(int(i) for i in str(m.cnn.kernel).split(','))
        stride = tuple(int(i) for i in str(m.cnn.stride).split(','))
        for i in range(m.cnn.layers):
            p = '{}/encoder/conv_{}'.format(name, i)
            specs += [(p + '/conv/kernel', (kernel[0], kernel[1], cin, m.num_filters), 'glorot'), (p + '/conv/bias', (m.num_filters,), 'zeros')]
            specs += ln(p, 0, m.num_filters)
            cin = m.num_filters
            f = -(-f // stride[1])
        din = f * m.num_filters
    specs.append(('{}/encoder/encoder_fc/kernel'.format(name), (din, de), 'glorot'))
    for i in range(m.encoder.num_blocks):
        p = '{}/encoder/block_{}'.format(name, i)
        specs += attn_self(p + '/encoder_self_attention', de) + ln(p, 0, de) + ffn(p, de) + ln(p, 1, de)
    specs.append(('{}/decoder/decoder/fully_connected/kernel'.format(name), (de, V), 'glorot'))
    d2 = de
    if m.use_semantic_encoder:
        d2 = m.encoder2.num_cell_units
        for i in range(m.encoder2.num_blocks):
            p = '{}/TPEncoder/block_{}'.format(name, i)
            specs += attn_self(p + '/encoder_self_attention', d2) + ln(p, 0, d2) + ffn(p, d2) + ln(p, 1, d2)
    dd = m.decoder.num_cell_units
    for i in range(m.decoder.num_blocks):
        p = '{}/decoder/block_{}'.format(name, i)
        specs += attn_self(p + '/decoder_self_attention', dd) + ln(p, 0, dd)
        specs += attn_cross(p + '/decoder_vanilla_attention', dd, dd, d2) + ln(p, 1, dd)
        specs += ffn(p, dd) + ln(p, 2, dd)
    specs.append(('{}/decoder/decoder_fc/kernel'.format(name), (dd, V), 'glorot'))
    if m.loss_type and 'kd' in m.loss_type:
        dk = m.kd.num_cell_units
        specs += attn_cross('{}/decoder_vanilla_attention'.format(name), dk, dk, d2)
    return specs


_STORES = {}
# the CTC branch behind the semantic encoder instead of beside it (see _encode); NEURST_DEFER_CTC=0: the round-1 order
_DEFER_CTC = os.environ.get('NEURST_DEFER_CTC', '1') != '0'


def get_variable_store(args, name, device):
    """Variables are shared by name between the train and the infer model (tf.AUTO_REUSE, :148,:496).  The store (five
    arenas of the parameter count) lives as long as the configuration object it was built for: when `args` is collected the
    entry goes with it, so a process that builds one model after another does not pile up arenas."""
    key = (id(args), name)
    ent = _STORES.get(key)
    if ent is None:
        dtype = ops.BF16 if str(args.compute_dtype).lower() in ('bf16', 'bfloat16') else ops.F32
        store = G.VariableStore(variable_specs(args, name), device, dtype, seed=1 if args.init_seed is None else args.init_seed)
        try:
            weakref.finalize(args, _STORES.pop, key, None)
            ent = _STORES[key] = (None, store)
        except TypeError:      # not weak-referenceable: keep `args` alive instead, a collected object could hand its id() on
            ent = _STORES[key] = (args, store)
    return ent[1]


_startup_frozen = False


def _freeze_startup_objects():
    """Once per process, before the first model is built: move what the interpreter holds so far (torch's and numpy's modules:
    ~170 k tracked objects) out of the cycle collector's reach (gc.freeze).  A training step leaves ~500 objects in reference
    cycles (tape closures), so a full collection comes round every ~35 steps, and each one walked those 170 k start-up
    objects: 25-60 ms during which nothing is enqueued - 0.5-1 ms per COSTT step on average (tools/exp/gc_series.py: chunks
    of ten steps at 8.6 ms with one of 10.5-14.7 ms among every six; flat 8.6 with the start-up objects frozen).  Models,
    batches and everything else created afterwards stay collectable.  NEURST_GC_FREEZE=0 leaves the collector alone."""
    global _startup_frozen
    if _startup_frozen or os.environ.get('NEURST_GC_FREEZE', '1') == '0':
        return
    _startup_frozen = True
    gc.collect()
    gc.freeze()


class TransformerTPE(Seq2SeqModel):
    """A general class for an encoder-decoder system with a two-pass encoder."""

    def __init__(self, tensor_global_step, encoder, decoder, is_train, args, summary=None, batch=None, embed_table_encoder=None,
                 embed_table_decoder=None, name='transformer', device=None):
        _freeze_startup_objects()
        self.args = args
        self.name = name
        self.ctc_decoder = args.model.decoder2.type
        self.mt_encoder = args.model.encoder2.type
        self.summary = summary
        self.size_embedding = args.model.decoder.size_embedding
        self.device = torch.device(device or ('cuda:%d' % torch.cuda.current_device()))
        self.store = get_variable_store(args, name, self.device)
        self.graph = G.Graph(self.store, self.store.compute_dtype, seed=args.seed or 0)
        self.embed_table_decoder = self.get_embedding(embed_table=None, size_input=args.dim_output, size_embedding=self.size_embedding)
        if args.dirs and args.dirs.src_vocab:
            raise NotImplementedError('a separate source vocabulary (dirs.src_vocab) is outside the hot path')
        self.src_embed_table = self.embed_table_decoder
        self.ctc_merge_repeated = args.model.decoder2.ctc_merge_repeated
        for flag in ('is_multilingual', 'is_cotrain', 'encoder_concat_tag', 'decoder_concat_tag', 'pt_decoder', 'variational_inference',
                     'semantic_encoder_pretrain', 'use_wrapctc'):
            if getattr(args.model, flag):
                raise NotImplementedError('model.{} is outside the hot path (False/absent in both templates)'.format(flag))
        if not self.ctc_merge_repeated:
            raise NotImplementedError('ctc_merge_repeated: False is outside the hot path (True in both templates)')
        self.loss_ast = self.loss_am = self.loss_bert = None
        self.last = {}
        super().__init__(tensor_global_step, encoder, decoder, is_train, args, batch, embed_table_encoder=None, embed_table_decoder=None, name=name)

    # ------------------------------------------------------------------ graph construction
    def build_graph(self):
        self.optimizer = self.build_optimizer()
        handle_gradients(self.args)
        if self.args.use_horovod:
            raise NotImplementedError('use_horovod: data parallelism is one process per GPU over RCCL (torch.distributed)')
        self.bucketer = GradBucketer(self.store, bucket_bytes=max(int(float(self.args.grad_bucket_mb or 64) * (1 << 20)) // 256 * 256, 256),
                                     producer_streams=lambda: (self.graph.wgrad_stream, self._ctc_stream), cuts=self._bucket_cuts())
        # shrink_layer (COSTT) without its mid-step read-back: NEURST_SHRINK_SPEC=0 restores the reference's behaviour
        self.graph.shrink_spec = None
        # (not with the KD branch: its attention over the shrunk sequence carries no key mask, so the padding counts there)
        if (self.args.model.shrink_layer and self.device.type == 'cuda' and self.args.model.loss_type != 'ctc_kd_ce'
                and os.environ.get('NEURST_SHRINK_SPEC', '1') != '0'):
            self.graph.shrink_spec = ShrinkSpeculation(self.device)
        if self.args.model.shrink_layer and self.graph.shrink_spec is None and G.side_wgrad_rows_from_env() is None:
            # The read-back that sizes shrink_layer's output drains the launch queue in mid-step; the small layers that
            # follow are then bound by the host's launch rate (measured: 13.9 of 14.4 ms/step), where the second stream's
            # extra host work only costs.  The large layers of the acoustic encoder's backward still gain from it.
            self.graph.side_wgrad_rows = (8192, 1 << 40)
        elif self.graph.shrink_spec is not None and 'NEURST_GROUP_MIN_ROWS' not in os.environ:
            # With the host running ahead, the decoder's and the semantic encoder's weight gradients (2560 and ~1024 rows, 66
            # launches of 20-40 us that occupy 36-144 CUs each) go out as whole-K tiles of the grouped kernel on the second
            # stream instead: 11.98 -> 10.36 ms/step from the second stream alone, 10.10 with the groups (LUT, whose decoder
            # has 1280 rows, measured slower with them: 13.15 -> 13.71, and keeps the default threshold between its encoder and decoder rows)
            self.graph.group_min_rows = 1024
        self.graph.on_param_grad = self.bucketer.param_done
        self.graph.on_node_done = self.bucketer.node_done
        self.frozen = [s for s in str(self.args.frozen_scope).split(',') if s] if self.args.frozen_scope else []
        self._make_modules()
        self.__class__.num_Instances += 1
        # [loss, shape_batch, op_optimize, debug] ordering of :193 as callables
        return [lambda: self.last.get('loss'), lambda: self.last.get('shape_batch'), self.train_step, lambda: None]

    def _bucket_cuts(self):
        """Arena offsets at which a gradient bucket must end (GradBucketer).  The arena is laid out in forward order and
        backward completes it from the end - except the decoder's embedding table, first in the arena but complete as soon
        as the decoder's backward is: it gets a bucket of its own.  Behind it sit the variables whose gradients complete
        LAST (encoder_fc, the first acoustic block): the first `grad_tail_mb` (default 8) MB of them form the last bucket, so
        that what is exchanged after backward has ended is a few MB and not a whole 64 MB bucket."""
        tail = int(float(self.args.grad_tail_mb or 8) * (1 << 20)) // 4
        pv = list(self.store.vars.values())                      # arena order
        cuts, start = [], 0
        if pv and pv[0].name.endswith('/embedding') and len(pv) > 1:
            start = pv[1].offset
            cuts.append(start)
        for p in pv:
            if p.offset >= start + tail:
                cuts.append(p.offset)
                break
        return cuts

    def build_infer_graph(self):
        self._make_modules()
        return [lambda *a: self.infer(*a)[0], lambda *a: self.infer(*a)[1], lambda: self.last.get('shape_batch'), lambda: None]

    def _make_modules(self):
        a = self.args
        self.encoder = self.gen_encoder(is_train=self.is_train, embed_table=None, args=a)
        self.encoder_mt = self.mt_encoder(is_train=self.is_train, embed_table=None, args=a, name='TPEncoder')
        self.decoder = self.gen_decoder(is_train=self.is_train, embed_table=self.embed_table_decoder, global_step=self.global_step, args=a)
        self.decoder_ctc = self.ctc_decoder(is_train=self.is_train, embed_table=self.src_embed_table, global_step=self.global_step, args=a)

    # ------------------------------------------------------------------ pieces of build_single_graph
    def ctc_loss(self, logits, len_logits, labels, len_labels, grad_scale=1.0, recursion_stream=None):
        """tf.nn.ctc_loss(..., ctc_merge_repeated=True, ignore_longer_outputs_than_inputs=True) (:93-102).
        Returns (loss [B], argmax [B,T]) and registers the gradient wrt logits on the tape."""
        g = self.graph
        loss, grad, argmax = ops.ctc_loss_fwd_bwd(logits, len_logits, labels, len_labels, g.dtype, grad_scale, in_place=(g.dtype == ops.F32),
                                                  recursion_stream=recursion_stream)
        if g.tape is not None:
            g.tape.set(logits, grad)
        return loss, argmax

    _ctc_stream = None
    _ctc_pending = False

    def _join_ctc_stream(self):
        if self._ctc_pending:
            torch.cuda.current_stream().wait_stream(self._ctc_stream)
            self._ctc_pending = False

    def ctc_decode(self, argmax, len_logits):
        """tf.nn.ctc_greedy_decoder(merge_repeated=True) + sparse_to_dense (:106-125, :251-259)."""
        if self.args.ctc_beam_size and self.args.ctc_beam_size > 1:
            raise NotImplementedError('ctc_beam_size > 1 is outside the hot path')
        return ops.ctc_greedy(argmax, len_logits, self.args.dim_output - 1)

    def _to_device(self, batch):
        out = {}
        for k, v in batch.items():
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(v)
            out[k] = v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) else v
        return out

    def _front_end(self, batch):
        """Step 0 (:217-228) plus, when the batch carries raw fbank frames, process_raw_feature (a1)."""
        a = self.args
        feat, len_fea = batch['feature'], batch['len_fea']
        if feat.shape[-1] != a.data.dim_input:
            assert feat.shape[-1] == a.data.dim_feature, 'feature width {} is neither dim_input nor dim_feature'.format(feat.shape[-1])
            feat, len_fea = tfRecord.process_raw_feature(feat, len_fea, a, out_dtype=ops.F32)
        self.last['specaug_params'] = None
        if self.is_train and a.model.use_specAug:
            if feat is batch['feature']:
                feat = feat.clone()    # the masks are applied in place; keep the caller's buffer intact
            feat, params = apply_specaugmant(feat, 30, 40, 2, 2, 0.2, key=self.graph.key(9000), params=batch.get('specaug_params'))
            self.last['specaug_params'] = params
        return feat, len_fea

    def _encode(self, feature, len_fea, want_ctc_grad, ctc_scale, batch):
        """Steps 1-3 (:229-284).  Returns encoded, len_encoded, logits_am, loss_ctc, argmax_am."""
        a = self.args
        g = self.graph
        encoded_am, len_encoded_am = self.encoder(features=feature, len_feas=len_fea)
        # The CTC branch (head GEMM, loss, its gradient, greedy decode) only meets the rest of the step again at the loss
        # mix and in the backward of the CTC head, so it runs on its own HIP stream - unless shrink_layer needs the arg-max
        # right away.  It is enqueued BEHIND the semantic encoder (the stream waits for everything the main stream holds
        # at that point): its kernels then run beside the decoder's chain of decoder-size launches instead of sharing the
        # chip with the semantic encoder's GEMMs, and the head GEMM leaves the main stream.  Measured 14.14 -> 14.06 ms
        # per step: kernels of two HIP streams hardly overlap on this part (tools/exp/cu_mask.py: a chain of 200 small
        # GEMMs beside 10 large ones takes the SUM of both, and confining the large ones to a CU-masked stream
        # (hipExtStreamCreateWithCUMask, 64-192 CUs) makes the pair 1.5-4 x slower than running them one after the other).
        side = None
        if want_ctc_grad and not a.model.shrink_layer and encoded_am.is_cuda and g.side_streams:
            if self._ctc_stream is None:
                self._ctc_stream = G.shared_stream('ctc', self.device)
            side = self._ctc_stream
        defer = side is not None and _DEFER_CTC and a.model.use_semantic_encoder
        if defer:
            encoded, len_encoded = self.encoder_mt(features=encoded_am, len_feas=len_encoded_am)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with G.variable_scope(self.decoder.name or 'decoder'):
                    if g.side_gemm_grid:            # see graph.py: the head GEMM leaves CUs to the decoder's chain
                        ops.set_option('gemm256_max_grid', g.side_gemm_grid)
                    try:
                        logits_am, _, len_logits_am = self.decoder_ctc(encoded=encoded_am, len_encoded=len_encoded_am)
                    finally:
                        if g.side_gemm_grid:
                            ops.set_option('gemm256_max_grid', 0)
                if g.dtype == ops.F32 and self.keep_logits:
                    self.last['logits_am'] = logits_am.clone()      # the fp32 gradient overwrites the logits in place
                loss_ctc, argmax_am = self.ctc_loss(logits=logits_am, len_logits=len_logits_am, labels=batch['aux_label'],
                                                    len_labels=batch['len_aux_label'], grad_scale=ctc_scale)
                preds_am, len_preds_am = self.ctc_decode(argmax_am, len_logits_am)
            self._ctc_pending = True
            self.last['preds_am'], self.last['len_preds_am'] = preds_am, len_preds_am
            return encoded, len_encoded, logits_am, loss_ctc
        with G.variable_scope(self.decoder.name or 'decoder'):
            logits_am, _, len_logits_am = self.decoder_ctc(encoded=encoded_am, len_encoded=len_encoded_am)
        loss_ctc = None
        if want_ctc_grad:
            if g.dtype == ops.F32 and self.keep_logits:
                self.last['logits_am'] = logits_am.clone()      # the fp32 gradient overwrites the logits in place
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    loss_ctc, argmax_am = self.ctc_loss(logits=logits_am, len_logits=len_logits_am, labels=batch['aux_label'],
                                                        len_labels=batch['len_aux_label'], grad_scale=ctc_scale)
                    preds_am, len_preds_am = self.ctc_decode(argmax_am, len_logits_am)
                self._ctc_pending = True
            else:
                # shrink_layer needs the arg-max now: only the recursion and the gradient correction leave the main stream
                rs = None
                if logits_am.is_cuda and g.side_streams:
                    if self._ctc_stream is None:
                        self._ctc_stream = G.shared_stream('ctc', self.device)
                    rs = self._ctc_stream
                loss_ctc, argmax_am = self.ctc_loss(logits=logits_am, len_logits=len_logits_am, labels=batch['aux_label'],
                                                    len_labels=batch['len_aux_label'], grad_scale=ctc_scale, recursion_stream=rs)
                self._ctc_pending = rs is not None
        else:
            B, T, V = logits_am.shape
            argmax_am = ops.argmax_rows(logits_am.view(B * T, V)).view(B, T)
            if self.keep_logits:
                self.last['logits_am'] = logits_am
        if side is None:
            preds_am, len_preds_am = self.ctc_decode(argmax_am, len_logits_am)
        self.last['preds_am'], self.last['len_preds_am'] = preds_am, len_preds_am
        if a.model.shrink_layer:
            encoded_am, len_encoded_am = shrink_layer(encoded_am, len_encoded_am, logits_am, encoded_am.shape[-1], argmax=argmax_am)
        if a.model.use_semantic_encoder:
            encoded, len_encoded = self.encoder_mt(features=encoded_am, len_feas=len_encoded_am)
        else:
            encoded, len_encoded = encoded_am, len_encoded_am
        return encoded, len_encoded, logits_am, loss_ctc

    def loss_weights(self):
        m = self.args.model
        lt = m.loss_type
        if lt == 'CE':
            return 1.0, 0.0, 0.0
        if lt == 'ce_ctc':                      # :385-398
            return 1 - m.decoder2.kd_factor, m.decoder2.kd_factor, 0.0
        if lt == 'ctc_kd_ce':                   # :400-418
            return 1 - m.decoder2.kd_factor - m.kd.kd_factor, m.decoder2.kd_factor, m.kd.kd_factor
        raise NotImplementedError('loss_type {} is outside the hot path (templates use ctc_kd_ce / ce_ctc)'.format(lt))

    keep_logits = False   # tests set this to read logits back

    def forward_backward(self, batch):
        """build_single_graph with is_train=True: losses + gradients of every variable (in store.grads)."""
        a = self.args
        g = self.graph
        batch = self._to_device(batch)
        tape = G.Tape()
        with G.use_graph(g), G.variable_scope(self.name):
            g.begin_forward(tape, step=self.global_step.value)
            g.dropout_enabled = True
            self.bucketer.reset()
            self.store.zero_grads()
            w_ast, w_ctc, w_kd = self.loss_weights()
            feature, len_fea = self._front_end(batch)
            B = feature.shape[0]
            self.last['shape_batch'] = tuple(feature.shape)
            encoded, len_encoded, logits_am, loss_ctc = self._encode(feature, len_fea, w_ctc != 0, w_ctc / B, batch)
            # Step 4: decoder, teacher forcing (:300-343)
            Tensors = namedtuple('tensors_input', 'label_splits, len_label_splits')
            with G.variable_scope(self.decoder.name or 'decoder'):
                decoder_input = self.decoder.build_input(id_gpu=0, tensors_input=Tensors([batch['label']], [batch['len_label']]))
                logits, _, _ = self.decoder.decode(encoded=encoded, len_encoded=len_encoded, decoder_input=decoder_input.input_labels)
            if g.dtype == ops.F32 and self.keep_logits:
                self.last['logits'] = logits.clone()
            elif self.keep_logits:
                self.last['logits'] = logits
            xent = self.ce_loss(logits=logits, labels=decoder_input.output_labels, len_labels=decoder_input.len_labels, grad_scale=w_ast / B)
            mse_partial, mse_numel = None, 0
            if w_kd:
                bert_feat = batch['bert_feat']
                encoded_feat = extract_feat(encoded, bert_feat, a)
                mse_partial = compute_kd_loss(bert_feat, encoded_feat, a.model.kd.loss_type, grad_scale=w_kd)
                mse_numel = encoded_feat.numel()
            # Nothing on the main stream needs the CTC branch before the loss mix: when the CTC head's backward runs as an
            # early node on the weight-gradient stream, that stream waits for the CTC stream and the main stream goes
            # straight into the decoder's backward; the loss mix follows the backward.
            lazy = self._ctc_pending and _DEFER_CTC and bool(tape.early) and g.side_streams
            if lazy:
                g.early_stream = self._ctc_stream
            else:
                self._join_ctc_stream()
                out4 = ops.loss_combine(xent, loss_ctc, mse_partial, mse_numel, B, xent.shape[1], w_ast, w_ctc, w_kd, self.device)
            self.last.update(encoded=encoded, len_encoded=len_encoded, loss_ctc_vec=loss_ctc, xent=xent)
            tape.backward()           # `optimizer.compute_gradients(loss)` (:472)
            g.early_stream = None
            g.join_side_streams()
            if lazy:
                self._join_ctc_stream()
                out4 = ops.loss_combine(xent, loss_ctc, mse_partial, mse_numel, B, xent.shape[1], w_ast, w_ctc, w_kd, self.device)
            g.tape = None
        return out4

    def frozen_ranges(self):
        """Arena ranges of the variables `frozen_scope` names (:171-183): the reference drops them from the list handed to
        apply_gradients AFTER handle_gradients, so their gradients still count in the global norm, and neither the
        variables nor their Adam slots ever change."""
        if not self.frozen:
            return []
        al = G.ALIGN      # a variable owns its padding too: the ranges Adam still sees start on 256-byte boundaries
        return [(p.offset, p.offset + -(-p.size // al) * al) for p in self.store.vars.values() if any(s in p.name for s in self.frozen)]

    def _step(self, batch, _replay=False):
        """forward + backward + gradient mean + Adam.  NEURST_OVERLAP_ADAM selects where the Adam kernel runs:
        `off`  (default) one launch over the whole arena after the gradient mean;
        `tail` per bucket, after backward, in the order the all-reduces were launched and on the communication stream -
               early buckets are updated while later ones are still in flight;
        `backward` per bucket as soon as backward has finished with it (GradBucketer).
        Measured on one MI355X with the bucket path forced on over RCCL (one rank, so no real traffic to hide behind):
        off 16.65, backward 16.67, tail 16.99 ms/step - 18 small Adam launches cost more than one large one, so the
        overlapped modes stay opt-in until a multi-GPU measurement says otherwise."""
        spec = self.graph.shrink_spec
        if spec is not None and not _replay:
            self._settle_shrink(wait=False)
        mode = os.environ.get('NEURST_OVERLAP_ADAM') or 'off'
        mode = {'0': 'off', '1': 'backward'}.get(mode, mode)
        overlap = mode in ('tail', 'backward') and (not self.frozen) and self.graph.side_streams and self.device.type == 'cuda' and spec is None
        if self.args.lambda_l2 or self.args.grad_clip_global_norm:
            overlap = False           # the L2 penalty and the global-norm clip need the complete averaged gradient before Adam
        self.bucketer.apply_in_backward = mode == 'backward'
        if overlap:
            lr, self.bucketer.apply_fn = self.optimizer_update_fn(self.store, 1.0 / self.bucketer.world)
        try:
            out4 = self.forward_backward(batch)
            scale = average_gradients(self.bucketer)
            # gradientTools.py:40-67, after the tower mean and BEFORE the frozen-scope filter (TransformerTPE.py:167-183)
            scale = handle_gradients(self.args, self.store, scale, produced=self.graph.produced)
        finally:
            self.bucketer.apply_fn = None
        if overlap:
            self.finish_update(self.store)
            self.last['lr'] = lr
        else:
            self.last['lr'] = self.apply_gradients(self.store, scale, skip=self.frozen_ranges(), skip_flag=None if spec is None else spec.state)
        if spec is not None:
            ev, host, bound = spec.last
            spec.pending.append((ev, host, bound, batch, self.global_step.value - 1))
        return out4

    def _settle_shrink(self, wait):
        """Look at the shrink results of the steps whose GPU work has finished (wait=True: of every enqueued step): feed the
        observed maxima to the bound heuristic, and repeat - with the full length - every batch whose kept frames did not
        fit its bound (its Adam update was vetoed on the device).  The host-side step counters are rolled back first (a
        vetoed step never happened): directly after a step whose losses were read back (train_step, nothing else enqueued)
        the repeat IS that step, bit for bit; in train_step_async the batch is simply processed a few steps late."""
        spec = self.graph.shrink_spec
        redo = []
        while spec.pending:
            ev, host, bound, batch, step = spec.pending[0]
            if wait:
                ev.synchronize()
            elif not ev.query():
                break
            spec.pending.popleft()
            overflow, seen = int(host[0]), int(host[1])
            spec.free.append((ev, host))
            spec.history.append(max(seen, 1))
            if overflow:
                redo.append((batch, step))
        out = None
        for batch, step in redo:
            self.global_step.value -= 1
            self.store.adam_t -= 1
            spec.force_full = True
            try:
                out = self._step(batch, _replay=True)
            finally:
                spec.force_full = False
            self.last['shrink_replays'] = self.last.get('shrink_replays', 0) + 1
        return out

    def train_step(self, batch):
        """One `sess.run([model.list_run, loss_ast, loss_am, loss_bert])` (run_lut.py:365-368):
        returns (loss, shape_batch, loss_ast, loss_ctc, loss_bert) as Python floats (one 16-byte read back)."""
        vals = self._step(batch).cpu().numpy()
        if self.graph.shrink_spec is not None:                  # the read-back above has waited for the GPU: settle now, exactly
            again = self._settle_shrink(wait=True)
            if again is not None:
                vals = again.cpu().numpy()
        self.last['loss'] = float(vals[0])
        self.loss_ast, self.loss_am, self.loss_bert = float(vals[1]), float(vals[2]), float(vals[3])
        return self.last['loss'], self.last['shape_batch'], self.loss_ast, self.loss_am, self.loss_bert

    def train_step_async(self, batch):
        """Same as train_step without the read back (throughput runs): returns the [4] device tensor."""
        return self._step(batch)

    # ------------------------------------------------------------------ inference (:486-513, :314-325)
    def infer(self, features, len_feas):
        """-> (preds [B, steps] int32 numpy, preds_am [B, T] int32 numpy zero padded)."""
        a = self.args
        if a.dirs and a.dirs.lm_checkpoint:
            raise NotImplementedError('LM rerank (dirs.lm_checkpoint) is outside the hot path')
        beam = (a.beam_size or 1) > 1
        g = self.graph
        batch = self._to_device({'feature': features, 'len_fea': len_feas})
        with G.use_graph(g), G.variable_scope(self.name):
            g.begin_forward(None)
            g.dropout_enabled = False
            feature, len_fea = self._front_end(batch)
            self.last['shape_batch'] = tuple(feature.shape)
            encoded, len_encoded, logits_am, _ = self._encode(feature, len_fea, False, 0.0, batch)
            self.last.update(encoded=encoded, len_encoded=len_encoded)
            with G.variable_scope(self.decoder.name or 'decoder'):
                if beam:      # TransformerTPE.py:314-319
                    self.last['beams'], preds, len_decoded = self.decoder.beam_decode_rerank(encoded, len_encoded)
                else:
                    _, preds, len_decoded = self.decoder.decoder_with_caching(encoded, len_encoded)
        self.last['len_decoded'] = len_decoded
        preds_am = self.last['preds_am'].cpu().numpy()
        maxlen = max(int(self.last['len_preds_am'].cpu().numpy().max()), 0)
        return preds, preds_am[:, :maxlen]
