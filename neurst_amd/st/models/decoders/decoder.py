"""Decoder base class, interface of st/models/decoders/decoder.py:11-204."""
from collections import namedtuple

from .. import graph as G
from ...tools.tftools.tfTools import right_shift_rows
from .... import ops


class Decoder(object):
    def __init__(self, args, is_train, global_step, embed_table=None, name=None):
        self.args = args
        self.name = name
        self.is_train = is_train
        self.start_token = args.token2idx['<sos>']
        self.end_token = args.token2idx['<eos>']
        self.embed_table = embed_table
        self.global_step = global_step
        self.dim_output = args.dim_output

    def __call__(self, encoded, len_encoded):
        with G.variable_scope(self.name or 'decoder'):
            logits, sample_id, len_decode = self._decode(encoded, len_encoded)
        return logits, sample_id, len_decode

    def build_input(self, id_gpu, tensors_input):
        """decoder.py:63-92: input = <sos> + labels[:-1], output = labels."""
        decoder_input = namedtuple('decoder_input', 'input_labels, output_labels, len_labels')
        assert self.start_token, self.end_token
        if tensors_input.label_splits:
            decoder_input.output_labels = tensors_input.label_splits[id_gpu]
            decoder_input.input_labels = right_shift_rows(p=tensors_input.label_splits[id_gpu], shift=1, pad=self.start_token,
                                                          lens=tensors_input.len_label_splits[id_gpu])
            decoder_input.len_labels = tensors_input.len_label_splits[id_gpu]
        else:
            decoder_input.output_labels = None
            decoder_input.input_labels = None
            decoder_input.len_labels = None
        return decoder_input

    def embedding(self, ids, dropout_rate=0.0, site=None, pos0=0, signal_len=None, out=None):
        """decoder.py:198-204 lookup fused with `x += x + timing` and the dropout that follow it in
        transformer_decoder.py:193-199."""
        g = G.current()
        table = self.embed_table
        rate, key = g.rate(dropout_rate), g.key(site)
        out = ops.embedding_fwd(ids, table.master, g.dtype, rate, key, pos0=pos0, signal_len=signal_len, out=out)
        tape = g.tape
        if tape is not None:
            def bwd():
                d = tape.pop(out)
                if d is not None:
                    ops.embedding_bwd(ids, d, table.grad, rate, key)
                    g.param_grad_done(table)
            tape.push(bwd)
        return out

    def _decode(self, encoded, len_encoded):
        raise NotImplementedError
