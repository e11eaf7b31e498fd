#!/usr/bin/env python
"""Timing-only ablations of the 256 x 256 kernel's K loop (gemm256_ablate: 1 no operand DMA, 4 no MFMA, 8 one barrier per phase,
16 no s_setprio pair, 64 DMA issued between the MFMAs; wrong results by construction) on a long-K NT problem."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neurst_amd import ops, lib
from tools.bench_kernels import timeit, DEV
L = lib.load()
rnd = lambda *s: (torch.rand(*s, device=DEV) * 2 - 1).to(torch.bfloat16)
L.neurst_set_option(b'gemm256_min_tiles', 1)
L.neurst_set_option(b'gemm256_bn', 256)
for (M, N, K) in ((8192, 8192, 8192), (16000, 768, 3072), (16000, 3072, 768)):
    A, B = rnd(M, K), rnd(N, K)
    C = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    for abl in (0, 1, 4, 8, 16, 64, 0):
        L.neurst_set_option(b'gemm256_ablate', abl)
        med, mn = timeit(lambda: ops.gemm(A, B, C, M, N, K, K, K, N, 0, 1), reps=12)
        print('M=%d N=%d K=%d ablate %2d: %8.1f us  (%6.1f TF if it were the real product)' % (M, N, K, abl, med, fl / med / 1e6), flush=True)
L.neurst_set_option(b'gemm256_ablate', 0)
