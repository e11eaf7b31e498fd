"""Dropout call-site ids: every dropout on the path has a fixed 32-bit site so the forward and
backward kernels (and the test oracle) regenerate the same mask.  Same table as oracle/model.py."""
SITE_ENC_IN, SITE_ENC2_IN, SITE_DEC_IN, SITE_KD_ATT = 0, 1, 2, 3
SITE_ENC_BLK, SITE_ENC2_BLK, SITE_DEC_BLK = 100, 300, 500


def blk_site(base, i, j):
    return base + 8 * i + j
