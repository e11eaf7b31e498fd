#!/usr/bin/env python
"""Square bf16 GEMMs (the shapes cdna_hip_programming.md quotes its 256^2 template on): this library's kernels against the vendor library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neurst_amd import ops, lib
from tools.bench_kernels import timeit, DEV
L = lib.load()
rnd = lambda *s: (torch.rand(*s, device=DEV) * 2 - 1).to(torch.bfloat16)
for n in (4096, 8192):
    for tb in (1, 0):
        A, B = rnd(n, n), rnd(n, n)
        C = torch.zeros(n, n, device=DEV, dtype=torch.bfloat16)
        fl = 2.0 * n ** 3
        line = 'n=%d %s ' % (n, 'NT' if tb else 'NN')
        for mt, bn, name in ((1 << 30, 0, '128-tile'), (1, 256, '256x256'), (1, 192, '256x192')):
            L.neurst_set_option(b'gemm256_min_tiles', mt)
            L.neurst_set_option(b'gemm256_bn', bn)
            med, mn = timeit(lambda: ops.gemm(A, B, C, n, n, n, n, n, n, 0, tb), reps=15)
            line += '| %s %7.1f us %6.1f TF ' % (name, med, fl / med / 1e6)
        Bt = B.t() if tb else B
        med, mn = timeit(lambda: torch.mm(A, Bt, out=C), reps=15)
        line += '| library %7.1f us %6.1f TF' % (med, fl / med / 1e6)
        print(line, flush=True)
