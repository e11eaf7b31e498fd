"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the dev-loop metrics.

* `wer`  follows st/tools/wer.py:124-151 line by line (numpy table of [substitutions, insertions, deletions], each
  cell the first candidate of smallest sum).
* `levenshtein` is the textbook distance that `editdistance.eval` (third-party, unpinned in requirements.txt, not
  importable here) returns for st/tools/textTools.py:34-70.
* `multi_bleu` follows st/tools/multi-bleu.perl:45-157 statement by statement (string-keyed n-gram hashes, closest
  reference length, my_log) and reproduces its printf line.

parity unpinned for BLEU beyond hand-derived cases: the reference holds no scored example (tests/test_metrics.py
derives a few by hand)."""
import math

import numpy as np


def wer(reference, decoded):
    errors = np.zeros([len(reference) + 1, len(decoded) + 1, 3])
    errors[0, :, 1] = np.arange(len(decoded) + 1)
    errors[:, 0, 2] = np.arange(len(reference) + 1)
    substitution = np.array([1, 0, 0])
    insertion = np.array([0, 1, 0])
    deletion = np.array([0, 0, 1])
    for r, ref in enumerate(reference):
        for d, dec in enumerate(decoded):
            errors[r + 1, d + 1] = min((errors[r, d] + (ref != dec) * substitution, errors[r + 1, d] + insertion,
                                        errors[r, d + 1] + deletion), key=np.sum)
    return tuple(errors[-1, -1])


def levenshtein(a, b):
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def multi_bleu(refs, hyp_lines, lowercase=False):
    """refs: list of corpora (lists of lines without the newline); returns the printed line"""
    CORRECT, TOTAL = {}, {}
    length_translation = length_reference = 0
    for s, line in enumerate(hyp_lines):
        if lowercase:
            line = line.lower()
        WORD = line.split()
        REF_NGRAM = {}
        closest_diff, closest_length = 9999, 9999
        for corpus in refs:
            if s >= len(corpus):
                continue
            reference = corpus[s].lower() if lowercase else corpus[s]
            RW = reference.split()
            length = len(RW)
            diff = abs(len(WORD) - length)
            if diff < closest_diff:
                closest_diff, closest_length = diff, length
            elif diff == closest_diff:
                if length < closest_length:
                    closest_length = length
            for n in range(1, 5):
                REF_NGRAM_N = {}
                for start in range(0, len(RW) - (n - 1)):
                    ngram = str(n) + ''.join(' ' + RW[start + w] for w in range(n))
                    REF_NGRAM_N[ngram] = REF_NGRAM_N.get(ngram, 0) + 1
                for ngram, c in REF_NGRAM_N.items():
                    if ngram not in REF_NGRAM or REF_NGRAM[ngram] < c:
                        REF_NGRAM[ngram] = c
        length_translation += len(WORD)
        length_reference += closest_length
        for n in range(1, 5):
            T_NGRAM = {}
            for start in range(0, len(WORD) - (n - 1)):
                ngram = str(n) + ''.join(' ' + WORD[start + w] for w in range(n))
                T_NGRAM[ngram] = T_NGRAM.get(ngram, 0) + 1
            for ngram, c in T_NGRAM.items():
                TOTAL[n] = TOTAL.get(n, 0) + c
                if ngram in REF_NGRAM:
                    CORRECT[n] = CORRECT.get(n, 0) + (c if REF_NGRAM[ngram] >= c else REF_NGRAM[ngram])
    bleu = {}
    for n in range(1, 5):
        bleu[n] = (CORRECT.get(n, 0) / TOTAL[n] if TOTAL[n] else 0) if n in TOTAL else 0
    if length_reference == 0:
        return 'BLEU = 0, 0/0/0/0 (BP=0, ratio=0, hyp_len=0, ref_len=0)'
    brevity_penalty = 1
    if length_translation < length_reference:
        brevity_penalty = math.exp(1 - length_reference / length_translation)
    my_log = lambda x: math.log(x) if x else -9999999999
    b = brevity_penalty * math.exp((my_log(bleu[1]) + my_log(bleu[2]) + my_log(bleu[3]) + my_log(bleu[4])) / 4)
    return 'BLEU = %.2f, %.1f/%.1f/%.1f/%.1f (BP=%.3f, ratio=%.3f, hyp_len=%d, ref_len=%d)' % (
        100 * b, 100 * bleu[1], 100 * bleu[2], 100 * bleu[3], 100 * bleu[4], brevity_penalty, length_translation / length_reference,
        length_translation, length_reference)
