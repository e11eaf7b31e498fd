"""st/tools/wer.py:18-151 - error rates with substitutions / insertions / deletions between two text files.
usage: python -m neurst_amd.st.tools.wer reference decoded"""
import sys

from .textTools import align_counts


def wer(reference, decoded):
    """:124-151 -> (substitutions, insertions, deletions) between two token lists"""
    return align_counts(reference, decoded)


def _rate(reference, decoded, tokens, name):
    substitutions = insertions = deletions = 0.0
    numwords = 0
    with open(reference, 'r') as fid, open(decoded, 'r') as did:
        for lref, lout in zip(fid, did):
            reftext, output = tokens(lref), tokens(lout)
            s, i, d = wer(reftext, output)
            substitutions += s
            insertions += i
            deletions += d
            numwords += len(reftext)
    substitutions /= numwords
    deletions /= numwords
    insertions /= numwords
    error = substitutions + deletions + insertions
    print('%s error rate: %s\n\tsubstitutions: %s\n\tinsertions: %s\n\tdeletions: %s' % (name, error, substitutions, insertions, deletions))
    return error


def calc_cer(reference, decoded):
    """:18-51 - characters, blanks removed"""
    return _rate(reference, decoded, lambda line: list(line.strip().replace(' ', '')), 'character')


def calc_wer(reference, decoded):
    """:54-85"""
    return _rate(reference, decoded, lambda line: line.strip().split(), 'word')


def calc_per(reference, decoded):
    """:88-121"""
    return _rate(reference, decoded, lambda line: line.strip().split(), 'phoneme')


if __name__ == '__main__':
    calc_wer(sys.argv[1], sys.argv[2])
