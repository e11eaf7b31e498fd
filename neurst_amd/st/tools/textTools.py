"""st/tools/textTools.py:14-123 - id arrays -> text and the per-batch error counts of the dev loop.  `editdistance.eval`
(a C extension the reference depends on, not in this image) is replaced by the library's `neurst_edit_distance`."""
import ctypes

import numpy as np

from ... import lib as _lib


def edit_distance(a, b):
    """Levenshtein distance between two sequences of hashable items (ids or words)."""
    if len(a) and not isinstance(a[0], (int, np.integer)) or len(b) and not isinstance(b[0], (int, np.integer)):
        table = {}
        a = [table.setdefault(x, len(table)) for x in a]
        b = [table.setdefault(x, len(table)) for x in b]
    a, b = np.ascontiguousarray(a, np.int32), np.ascontiguousarray(b, np.int32)
    d = _lib.load().neurst_edit_distance(a.ctypes.data, len(a), b.ctypes.data, len(b), None)
    if d < 0:
        _lib.check(int(d), 'edit_distance')
    return int(d)


def align_counts(reference, decoded):
    """(substitutions, insertions, deletions) as st/tools/wer.py:124-151 counts them"""
    table = {}
    a = np.ascontiguousarray([table.setdefault(x, len(table)) for x in reference], np.int32)
    b = np.ascontiguousarray([table.setdefault(x, len(table)) for x in decoded], np.int32)
    sid = (ctypes.c_int64 * 3)()
    d = _lib.load().neurst_edit_distance(a.ctypes.data, len(a), b.ctypes.data, len(b), sid)
    if d < 0:
        _lib.check(int(d), 'edit_distance')
    return float(sid[0]), float(sid[1]), float(sid[2])


def unpadding(list_idx, eos_idx=None, min_idx=0, max_idx=None):
    """:14-31 - the ids before the first `eos_idx` that lie strictly between min_idx and max_idx (1-d array)"""
    ids = np.asarray(list_idx)
    if eos_idx is not None:
        hits = np.flatnonzero(ids == eos_idx)
        if hits.size:
            ids = ids[:hits[0]]
    keep = ids > min_idx
    if max_idx is not None:
        keep &= ids < max_idx
    return ids[keep]


def _error_counts(pairs):
    """sum of edit distances and of reference lengths over (result, reference) sequence pairs"""
    dist = length = 0
    for res, ref in pairs:
        dist += edit_distance(res, ref)
        length += len(ref)
    return dist, length


def batch_cer(result, reference, eos_idx=None, min_idx=0, max_idx=None):
    """:34-47 -> (sum of edit distances over token ids, sum of reference lengths)"""
    clean = lambda row: unpadding(row, eos_idx, min_idx, max_idx)
    return _error_counts((clean(res), clean(ref)) for res, ref in zip(result, reference))


def batch_wer(result, reference, idx2token, unit, eos_idx=None, min_idx=0, max_idx=None, args=None):
    """:50-70 -> the same over the whitespace-separated words of the detokenised text"""
    words = lambda row: array2text(row, unit, idx2token, eos_idx, min_idx, max_idx, args=args).split()
    return _error_counts((words(res), words(ref)) for res, ref in zip(result, reference))


def array2text(res, unit, idx2token, eos_idx=None, min_idx=0, max_idx=None, args=None, delimiter=None):
    """:73-93"""
    res = unpadding(res, eos_idx, min_idx, max_idx)
    if unit == 'char':
        return array_idx2char(res, idx2token, seperator='')
    if unit == 'word':
        return array_idx2char(res, idx2token, seperator=' ')
    if unit == 'subword':
        text = array_idx2char(res, idx2token, seperator=' ')
        if delimiter:
            return text.replace(delimiter, '')
        if args and args.data.delimiter:
            return text.replace(args.data.delimiter, '')
        return text.replace('@@ ', '')
    raise NotImplementedError('not know unit!')


def array_idx2char(array_idx, idx2token, seperator=''):
    """:119-123"""
    if len(array_idx) == 0 or np.isscalar(array_idx[0]):
        return seperator.join(idx2token[int(i)] for i in array_idx)
    return [array_idx2char(i, idx2token, seperator=seperator) for i in array_idx]
