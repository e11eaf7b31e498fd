"""CTC head, st/models/decoders/fc_decoder.py:13-42: bias-free linear d -> V ([d, V] kernel).
The arg-max the reference returns as `preds` is produced by the CTC kernels (or by
ops.argmax_rows at inference), so it is returned as None here."""
import os

from .decoder import Decoder
from ..tools.utils import tf_layers_dense
from .... import ops


class FCDecoder(Decoder):
    def _decode(self, encoded, len_encoded):
        dim_output = self.args.src_dim_output if self.args.src_dim_output else self.args.dim_output
        # the gradient of these logits comes straight from the CTC loss: the head's backward need not wait for the decoder's
        logits = tf_layers_dense(inputs=encoded, units=dim_output, use_bias=False, name='fully_connected', out_dtype=ops.F32,
                                 early_backward=os.environ.get('NEURST_EARLY_CTC_BACKWARD', '1') != '0')
        return logits, None, len_encoded
