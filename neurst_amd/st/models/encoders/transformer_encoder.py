"""Acoustic encoder, st/models/encoders/transformer_encoder.py:60-153: optional conv2d+LN+ReLU
sub-sampler and 3x3 max-pool (:69-103), key-padding bias, timing signal + dropout, `encoder_fc`
(Din -> d, no bias), num_blocks x {self-attention, residual+LN, FFN, residual+LN}, padded
frames zeroed (:151, fused into the last LayerNorm kernel)."""
import torch

from .encoder import Encoder
from .. import graph as G
from ..sites import SITE_ENC_IN, SITE_ENC_BLK, blk_site
from ..tools.utils import residual, multihead_attention, ff_hidden, tf_layers_dense
from ...layers import common_attention_v1, common_layers_v1
from .... import ops


def transformer_blocks(x, bias, len_feas, num_blocks, hidden_units, num_heads, attention_dropout_rate, residual_dropout_rate, site_base):
    for i in range(num_blocks):
        with G.variable_scope('block_{}'.format(i)):
            x = residual(x,
                         multihead_attention(query_antecedent=x, memory_antecedent=None, bias=bias,
                                             total_key_depth=hidden_units, total_value_depth=hidden_units,
                                             output_depth=hidden_units, num_heads=num_heads,
                                             dropout_rate=attention_dropout_rate, name='encoder_self_attention',
                                             summaries=True, site=blk_site(site_base, i, 0)),
                         dropout_rate=residual_dropout_rate, site=blk_site(site_base, i, 1))
            x = residual(x,
                         ff_hidden(inputs=x, hidden_size=4 * hidden_units, output_size=hidden_units, activation='relu'),
                         dropout_rate=residual_dropout_rate, site=blk_site(site_base, i, 2),
                         mask_lens=len_feas if i == num_blocks - 1 else None)   # "Mask padding part to zeros"
    return x


class Transformer_Encoder(Encoder):

    @staticmethod
    def normal_conv(inputs, filter_num, kernel, stride, padding, use_relu, name, w_initializer=None, norm_type='layer',
                    need_input_grad=True):
        """conv2d(SAME) -> layer_norm -> relu as one kernel (transformer_encoder.py:18-32)."""
        assert padding == 'SAME' and norm_type == 'layer' and use_relu
        g = G.current()
        with g.variable_scope(name):
            with g.variable_scope('conv'):
                w = g.get_variable('kernel', [kernel[0], kernel[1], inputs.shape[-1], filter_num])
                b = g.get_variable('bias', [filter_num])
            scale, bias = common_layers_v1.layer_norm_vars(filter_num)
        out, conv_out, mean, rstd = ops.conv2d_ln_relu_fwd(inputs, w.master, b.master, scale.master, bias.master, kernel, stride)
        tape = g.tape
        if tape is not None:
            def bwd():
                dout = tape.pop(out)
                if dout is None:
                    return
                dx = ops.conv2d_ln_relu_bwd(dout, inputs, w.master, scale.master, conv_out, out, mean, rstd, w.grad, b.grad,
                                            scale.grad, bias.grad, kernel, stride, need_input_grad)
                for p in (w, b, scale, bias):
                    g.param_grad_done(p)
                if need_input_grad:
                    tape.accumulate(inputs, dx)
            tape.push(bwd)
        return out

    def encode(self, features, len_feas):
        args = self.args
        g = G.current()
        attention_dropout_rate = args.model.encoder.attention_dropout_rate if self.is_train else 0.0
        residual_dropout_rate = args.model.encoder.residual_dropout_rate if self.is_train else 0.0
        hidden_units = args.model.encoder.num_cell_units
        num_heads = args.model.encoder.num_heads
        num_blocks = args.model.encoder.num_blocks
        has_cnn = bool(args.model.use_cnn_resample)
        if has_cnn:   # CNN sampling (:69-103)
            B, length = features.shape[0], features.shape[1]
            feat_size = int(args.data.dim_input / args.data.feature_channel)
            x = features
            if x.dtype != ops.torch_dtype(g.dtype):
                xb = torch.empty(x.shape, dtype=ops.torch_dtype(g.dtype), device=x.device)
                ops.cast_f32_to_bf16(x, xb)
                x = xb
            encoder_input = x.view(B, length, feat_size, args.data.feature_channel)
            kernel = tuple(int(i) for i in str(args.model.cnn.kernel).split(','))
            stride = tuple(int(i) for i in str(args.model.cnn.stride).split(','))
            for i in range(args.model.cnn.layers):
                encoder_input = self.normal_conv(inputs=encoder_input, filter_num=args.model.num_filters, kernel=kernel, stride=stride,
                                                 padding='SAME', use_relu=True, name='conv_' + str(i), norm_type='layer',
                                                 need_input_grad=i > 0)
                len_feas = ops.lens_ceil_div(len_feas, stride[0])
            pooled, parg = ops.maxpool3x3_fwd(encoder_input)
            tape = g.tape
            if tape is not None:
                src = encoder_input

                def bwd():
                    d = tape.pop(pooled)
                    if d is not None:
                        tape.accumulate(src, ops.maxpool3x3_bwd(d.view(pooled.shape), parg))
                tape.push(bwd)
            features3 = pooled.view(B, pooled.shape[1], -1)
            if tape is not None:   # the reshape shares the buffer: route its gradient to the pooled tensor
                def bwd_view():
                    d = tape.pop(features3)
                    if d is not None:
                        tape.set(pooled, d)
                tape.push(bwd_view)
            features = features3
        bias = common_attention_v1.attention_bias_ignore_padding(len_feas)
        x = common_attention_v1.add_timing_signal_1d(features, dropout_rate=residual_dropout_rate, site=SITE_ENC_IN,
                                                     stop_gradient=not has_cnn)
        x = tf_layers_dense(x, hidden_units, use_bias=False, name='encoder_fc', stop_gradient=not has_cnn)
        x = transformer_blocks(x, bias, len_feas, num_blocks, hidden_units, num_heads, attention_dropout_rate, residual_dropout_rate,
                               SITE_ENC_BLK)
        return x, len_feas
