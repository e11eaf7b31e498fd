"""Corpus BLEU as st/tools/multi-bleu.perl computes it (the dev loop shells out to that script,
st/tools/performanceTools.py:87-100): clipped n-gram precisions for n = 1..4 over whitespace tokens, the closest
reference length per sentence (the shorter of two equally close), brevity penalty exp(1 - ref/hyp), geometric mean with
log(0) = -9999999999, and the script's own output line."""
import math
from collections import Counter


def _ngrams(words, n):
    return Counter(tuple(words[i:i + n]) for i in range(len(words) - n + 1))


def multi_bleu(references, hypotheses, lowercase=False):
    """references: list of reference corpora (each a list of lines) or one list of lines; hypotheses: list of lines.
    Returns (bleu in [0, 100], the line multi-bleu.perl prints)."""
    if references and isinstance(references[0], str):
        references = [references]
    correct, total = [0] * 5, [0] * 5
    seen = [False] * 5                       # the script's defined($TOTAL[$n])
    len_hyp = len_ref = 0
    for s, line in enumerate(hypotheses):
        line = line.rstrip('\n')
        if lowercase:
            line = line.lower()
        words = line.split()
        closest_diff, closest_len = 9999, 9999
        ref_ngram = Counter()
        for corpus in references:
            if s >= len(corpus):
                continue
            ref = corpus[s].rstrip('\n')
            if lowercase:
                ref = ref.lower()
            rwords = ref.split()
            diff = abs(len(words) - len(rwords))
            if diff < closest_diff:
                closest_diff, closest_len = diff, len(rwords)
            elif diff == closest_diff and len(rwords) < closest_len:
                closest_len = len(rwords)
            for n in range(1, 5):
                for gram, c in _ngrams(rwords, n).items():
                    if ref_ngram[gram] < c:
                        ref_ngram[gram] = c
        len_hyp += len(words)
        len_ref += closest_len
        for n in range(1, 5):
            for gram, c in _ngrams(words, n).items():
                seen[n] = True
                total[n] += c
                correct[n] += min(c, ref_ngram.get(gram, 0))
    prec = [0.0] * 5
    for n in range(1, 5):
        prec[n] = correct[n] / total[n] if seen[n] and total[n] else 0.0
    if len_ref == 0:
        return 0.0, 'BLEU = 0, 0/0/0/0 (BP=0, ratio=0, hyp_len=0, ref_len=0)'
    bp = math.exp(1 - len_ref / len_hyp) if len_hyp < len_ref else 1.0
    log = lambda x: math.log(x) if x else -9999999999.0
    bleu = bp * math.exp(sum(log(prec[n]) for n in range(1, 5)) / 4)
    text = 'BLEU = %.2f, %.1f/%.1f/%.1f/%.1f (BP=%.3f, ratio=%.3f, hyp_len=%d, ref_len=%d)' % (
        100 * bleu, 100 * prec[1], 100 * prec[2], 100 * prec[3], 100 * prec[4], bp, len_hyp / len_ref, len_hyp, len_ref)
    return 100 * bleu, text


def bleu_from_files(ref_path, hyp_path, lowercase=False):
    """what `perl multi-bleu.perl ref < hyp | awk '{print($3)}' | awk -F, '{print $1}'` yields: the %.2f figure"""
    with open(ref_path) as fr, open(hyp_path) as fh:
        _, text = multi_bleu([fr.read().split('\n')[:-1]], fh.read().split('\n')[:-1], lowercase)
    return float(text.split()[2].rstrip(','))
