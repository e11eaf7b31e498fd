"""Encoder base class, interface of st/models/encoders/encoder.py:9-89."""
from .. import graph as G


class Encoder(object):
    def __init__(self, args, is_train, embed_table=None, name=None):
        self.args = args
        self.name = name
        self.is_train = is_train
        self.embed_table = embed_table

    def __call__(self, features, len_feas):
        with G.variable_scope(self.name or 'encoder'):
            outputs, output_seq_length = self.encode(features, len_feas)
        return outputs, output_seq_length

    def encode(self, features, len_feas):
        raise NotImplementedError
