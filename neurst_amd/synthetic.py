"""Synthetic batches of the shape the reference's TFRecord readers produce
(SURVEY.md 8d "Synthetic inputs").  The reference takes vocabularies, labels and BERT
features from data files that are not in its repository; ids here are a convention:
<pad>=0, <unk>=1, <sos>=2, <eos>=3, ..., <blk>=V-1 (st/tools/vocab.py appends <blk> last).
"""
import numpy as np


def make_batch(B, Traw, V, L, U=None, Lb=None, bert_dim=None, dim_raw=80, seed=1234, ragged=False,
               dtype=np.float32, eos=3, raw=True, dim_input=None):
    """Returns a dict with the fields of `build_tf_input_ctcbert` / `_multilabel`
    (st/models/lstmModel.py:290-354).  `feature` is RAW fbank [B, Traw, dim_raw] when
    raw=True (to be spliced/down-sampled, a1) or already [B, Traw, dim_input] otherwise."""
    rng = np.random.default_rng(seed)
    U = U or L
    width = dim_raw if raw else dim_input
    feat = rng.standard_normal((B, Traw, width)).astype(dtype)
    if ragged:
        len_fea = rng.integers(-(-Traw // 2), Traw + 1, size=B).astype(np.int32)
        len_fea[0] = Traw
    else:
        len_fea = np.full(B, Traw, np.int32)
    feat *= (np.arange(Traw)[None, :] < len_fea[:, None])[..., None]

    def labels(n, maxlen):
        lens = rng.integers(max(maxlen // 2, 1), maxlen + 1, size=n).astype(np.int32) if ragged else np.full(n, maxlen, np.int32)
        lens[0] = maxlen
        lab = rng.integers(4, max(V - 1, 5), size=(n, maxlen)).astype(np.int32)
        for b in range(n):
            lab[b, lens[b] - 1] = eos
            lab[b, lens[b]:] = 0
        return lab, lens

    label, len_label = labels(B, L)
    aux, len_aux = labels(B, U)
    batch = {'feature': feat, 'len_fea': len_fea, 'label': label, 'len_label': len_label,
             'aux_label': aux, 'len_aux_label': len_aux}
    if bert_dim:
        Lb = Lb or L
        batch['bert_feat'] = rng.standard_normal((B, Lb, bert_dim)).astype(dtype)
        batch['len_bert_feat'] = np.full(B, Lb, np.int32)
    return batch
