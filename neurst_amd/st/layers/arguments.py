"""Config loader with the reference's semantics (st/layers/arguments.py:19-89,
st/layers/utils.py:1557-1590): YAML -> nested AttrDict where a missing key reads as
None, several config files merged recursively, and the derived fields the hot path uses.
The reference's import-time side effects (directory creation, vocab / CSV / tfdata
loading, reading sys.argv) are NOT reproduced: vocabulary size and token ids are passed in.
"""
import os
import yaml


class AttrDict(dict):
    """Dictionary whose keys can be accessed as attributes; missing keys give None
    (st/layers/arguments.py:19-41)."""

    def __getattr__(self, item):
        try:
            if type(self[item]) is dict:
                self[item] = AttrDict(self[item])
            res = self[item]
        except Exception:
            res = None
        return res

    def __setattr__(self, key, value):
        self[key] = value


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def load_config(config_files):
    """Later files override earlier ones, recursively (st/layers/utils.py:1557-1590)."""
    if isinstance(config_files, str):
        config_files = [config_files]
    cfg = {}
    for path in config_files:
        with open(path, encoding='utf-8') as f:
            _merge(cfg, yaml.safe_load(f) or {})
    return cfg


def build_args(config, dim_output, token2idx=None, dim_feature=None, compute_dtype=None, seed=0, overrides=None):
    """`config`: path(s) of YAML file(s) or an already loaded dict.  Returns the `args`
    AttrDict the model classes read, with the derived fields of arguments.py:48-49,77-100,
    111-121,216-272,304-307."""
    cfg = config if isinstance(config, dict) else load_config(config)
    if overrides:
        _merge(cfg, overrides)
    args = AttrDict(cfg)
    args.num_gpus = len(str(args.gpus).split(',')) if args.gpus is not None else 1
    # The reference builds one batch for all its in-graph towers and tf.splits it (arguments.py:86-87, lstmModel.py:342-350).
    # Here every process is ONE tower that reads whole batches of its own (run_lut._Shard), so under a multi-process launch
    # (WORLD_SIZE set by torchrun) the per-tower size applies: a reference-style `gpus: '0,...,7'` would otherwise give
    # every rank an 8x batch, i.e. a 64x global batch.
    args.towers_per_batch = 1 if int(os.environ.get('WORLD_SIZE', '1')) > 1 else args.num_gpus
    args.list_gpus = ['/gpu:{}'.format(i) for i in range(args.num_gpus)]
    if args.bucket_boundaries:
        args.list_bucket_boundaries = [int(i) for i in str(args.bucket_boundaries).split(',')]
        args.list_batch_size = [int(args.num_batch_tokens / b) * args.towers_per_batch for b in args.list_bucket_boundaries] + [args.towers_per_batch]
    args.token2idx = token2idx or {'<pad>': 0, '<unk>': 1, '<sos>': 2, '<eos>': 3, '<blk>': dim_output - 1}
    args.dim_output = dim_output
    args.src_dim_output = dim_output
    args.eos_idx = args.token2idx.get('<eos>')
    args.sos_idx = args.token2idx.get('<sos>', args.token2idx.get('<blk>'))
    data = args.data
    data.dim_feature = dim_feature or data.dim_feature or data.dim_raw_input
    if dim_feature is None and data.dim_raw_input:
        data.dim_feature = data.dim_raw_input
    data.dim_input = data.dim_feature * ((data.num_context or 0) + 1) * (3 if data.add_delta else 1)
    if not data.feature_channel:
        data.feature_channel = 1
    # type strings -> classes (arguments.py:216-272)
    from ..models.encoders.transformer_encoder import Transformer_Encoder
    from ..models.encoders.transformer_TPEencoder import Transformer_TPEncoder
    from ..models.decoders.fc_decoder import FCDecoder
    from ..models.decoders.transformer_decoder import Transformer_Decoder
    from ..models.TransformerTPE import TransformerTPE
    table = {'transformer_encoder': Transformer_Encoder, 'transformer_tpeencoder': Transformer_TPEncoder,
             'fc_decoder': FCDecoder, 'transformer_decoder': Transformer_Decoder}
    m = args.model
    for part in ('encoder', 'encoder2', 'decoder', 'decoder2'):
        sub = getattr(m, part)
        if sub is not None and isinstance(sub.type, str):
            if sub.type not in table:
                raise NotImplementedError('not found {} type: {}'.format(part, sub.type))
            sub.type = table[sub.type]
    if m.structure != 'transformerTpe':
        raise NotImplementedError('only model.structure == transformerTpe is on the hot path (got {})'.format(m.structure))
    args.Model = TransformerTPE
    # build-side extensions (absent from the reference schema)
    args.compute_dtype = compute_dtype or args.compute_dtype or 'bf16'      # an explicit argument wins over the YAML key
    args.seed = args.seed if args.seed is not None else seed
    return args
