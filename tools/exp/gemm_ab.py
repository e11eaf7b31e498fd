#!/usr/bin/env python
"""A/B of the step's GEMM shapes: default kernel choice against the 256-tile kernel forced (gemm256_min_tiles = 1) and its two tile widths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools.bench_kernels as bk
from neurst_amd import lib
L = lib.load()
for mt, bn in ((128, 0), (1, 0), (1, 192), (1, 256)):
    L.neurst_set_option(b'gemm256_min_tiles', mt)
    L.neurst_set_option(b'gemm256_bn', bn)
    print('--- gemm256_min_tiles', mt, 'bn', bn, flush=True)
    bk.bench_gemm()
