"""Seq2SeqModel of st/models/seq2seqModel.py: ce_loss (:157-191) and get_embedding (:345-351)."""
from . import graph as G
from .lstmModel import LSTM_Model
from .model_tools import smoothing_cross_entropy


class Seq2SeqModel(LSTM_Model):
    def __init__(self, tensor_global_step, encoder, decoder, is_train, args, batch=None, embed_table_encoder=None,
                 embed_table_decoder=None, name='seq2seqModel'):
        self.gen_encoder = encoder
        self.gen_decoder = decoder
        self.embed_table_encoder = embed_table_encoder
        self.name = name
        super().__init__(tensor_global_step, is_train, args, batch=batch, name=name)

    def ce_loss(self, logits, labels, len_labels, grad_scale=1.0):
        """Label-smoothed CE, masked by length, sum / len per utterance (:162-189).  Returns the per-position
        terms xent [B,L] whose row sums are the reference's loss [B]; iter_mask is outside the hot path."""
        if self.args.model.iter_mask:
            raise NotImplementedError('model.iter_mask is outside the hot path')
        return smoothing_cross_entropy(logits=logits, labels=labels, vocab_size=self.args.dim_output,
                                       confidence=self.args.model.decoder.label_smoothing_confidence,
                                       len_labels=len_labels, grad_scale=grad_scale)

    def get_embedding(self, embed_table, size_input, size_embedding, name='embedding'):
        """seq2seqModel.py:345-351: variable `<model name>/<name>` [size_input, size_embedding]."""
        if size_embedding and embed_table is None:
            return self.store.get('{}/{}'.format(self.name, name), [size_input, size_embedding])
        return embed_table
