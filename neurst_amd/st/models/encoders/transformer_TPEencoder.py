"""Semantic ("TPE") encoder, st/models/encoders/transformer_TPEencoder.py:63-112: same blocks as
the acoustic encoder without conv / input projection; reads args.model.encoder2.*."""
from .encoder import Encoder
from .transformer_encoder import transformer_blocks
from ..sites import SITE_ENC2_IN, SITE_ENC2_BLK
from ...layers import common_attention_v1


class Transformer_TPEncoder(Encoder):
    def __init__(self, args, is_train, embed_table=None, name=None):
        self.name = name
        super().__init__(args, is_train, embed_table=embed_table, name=self.name)

    def encode(self, features, len_feas):
        cfg = self.args.model.encoder2
        attention_dropout_rate = cfg.attention_dropout_rate if self.is_train else 0.0
        residual_dropout_rate = cfg.residual_dropout_rate if self.is_train else 0.0
        bias = common_attention_v1.attention_bias_ignore_padding(len_feas)
        x = common_attention_v1.add_timing_signal_1d(features, dropout_rate=residual_dropout_rate, site=SITE_ENC2_IN)
        x = transformer_blocks(x, bias, len_feas, cfg.num_blocks, cfg.num_cell_units, cfg.num_heads, attention_dropout_rate,
                               residual_dropout_rate, SITE_ENC2_BLK)
        return x, len_feas
