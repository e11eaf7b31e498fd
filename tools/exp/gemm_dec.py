#!/usr/bin/env python
"""Decoder-size GEMMs of the LUT step (M = B L = 1280 rows): this library's choice against the vendor library (yardstick only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neurst_amd import ops
from tools.bench_kernels import timeit, DEV
rnd = lambda *s: (torch.rand(*s, device=DEV) * 2 - 1).to(torch.bfloat16)
M = int(os.environ.get('M', '1280'))
shapes = [('qkv fwd NT', 2304, 768, 1), ('out/q fwd NT', 768, 768, 1), ('ffn1 fwd NT', 3072, 768, 1), ('ffn2 fwd NT', 768, 3072, 1), ('fc fwd NT', 10000, 768, 1),
          ('qkv dgrad NN', 768, 2304, 0), ('out dgrad NN', 768, 768, 0), ('ffn1 dgrad NN', 768, 3072, 0), ('ffn2 dgrad NN', 3072, 768, 0), ('fc dgrad NN', 768, 10000, 0)]
for name, N, K, tb in shapes:
    A = rnd(M, K)
    B = rnd(N, K) if tb else rnd(K, N)
    C = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    fn = lambda: ops.gemm(A, B, C, M, N, K, K, B.shape[1], N, 0, tb)
    med, mn = timeit(fn, reps=30)
    lab = ops.gemm_kernel_label(A, C, M, N, K, 0, tb)
    Bt = B.t() if tb else B
    lm, _ = timeit(lambda: torch.mm(A, Bt, out=C), reps=30)
    fl = 2.0 * M * N * K
    print('%-14s M=%d N=%5d K=%5d  %6.1f us (min %5.1f) %6.1f TF  %-52s | library %6.1f us %6.1f TF' % (name, M, N, K, med, mn, fl / med / 1e6, lab, lm, fl / lm / 1e6), flush=True)
