"""KD head of st/models/kd_layer.py: extract_feat 'self_attention' variant (:33-47,:63-64) = cross-
attention with Q = BERT features, K/V = semantic-encoder output, bias=None, dropout
kd.attention_dropout_rate regardless of is_train; compute_kd_loss MSE over ALL elements (:9-11)."""
import torch

from . import graph as G
from .sites import SITE_KD_ATT
from .tools.utils import multihead_attention
from ... import ops


def make_attention(Q, V, args=None):
    kd = args.model.kd
    return multihead_attention(query_antecedent=Q, memory_antecedent=V, bias=None, total_key_depth=kd.num_cell_units,
                               total_value_depth=kd.num_cell_units, output_depth=kd.num_cell_units, num_heads=kd.num_heads,
                               dropout_rate=kd.attention_dropout_rate, name='decoder_vanilla_attention', summaries=True,
                               site=SITE_KD_ATT)


def extract_feat(encoded, bert_feat, args):
    type = args.model.kd.feat_extract_type
    if type != 'self_attention':
        raise NotImplementedError('only kd.feat_extract_type == self_attention is on the hot path (lut.yaml:74)')
    g = G.current()
    q = bert_feat
    if q.dtype != ops.torch_dtype(g.dtype):   # BERT features arrive as fp32
        qc = torch.empty(q.shape, dtype=ops.torch_dtype(g.dtype), device=q.device)
        ops.cast_f32_to_bf16(q, qc)
        q = qc
    return make_attention(q, encoded, args)


def compute_kd_loss(bert_feat, encoded, loss_type, grad_scale):
    """Returns the per-block partial sums of squared differences (reduced by neurst_loss_combine) and
    registers d(loss)/d(encoded) = grad_scale * 2 (encoded - bert) / numel on the tape."""
    if loss_type != 'mean_squared_error':
        raise NotImplementedError('only kd.loss_type == mean_squared_error is on the hot path (lut.yaml:75)')
    g = G.current()
    partial, dpred = ops.mse_fwd_bwd(encoded, bert_feat, grad_scale)
    if g.tape is not None:
        g.tape.set(encoded, dpred)
    return partial
