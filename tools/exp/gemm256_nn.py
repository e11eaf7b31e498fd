#!/usr/bin/env python
"""Data-gradient (NN) and forward (NT) shapes of the step: default routing against the 256-row kernel forced, 192- and 256-column tiles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neurst_amd import ops  # noqa: E402
from tools.bench_kernels import timeit, DEV  # noqa: E402

rnd = lambda *s: (torch.rand(*s, device=DEV) * 2 - 1).to(torch.bfloat16)
shapes = [
    ('dgrad ffn2 NN', 16000, 3072, 768, 0, 0),
    ('dgrad ffn1 NN', 16000, 768, 3072, 0, 0),
    ('dgrad qkv  NN', 16000, 768, 2304, 0, 0),
    ('dgrad out  NN', 16000, 768, 768, 0, 0),
    ('dgrad kv   NN', 16000, 768, 1536, 0, 0),
    ('fwd out    NT', 16000, 768, 768, 0, 1),
    ('fwd qkv    NT', 16000, 2304, 768, 0, 1),
    ('fwd ffn2   NT', 16000, 768, 3072, 0, 1),
]
for name, M, N, K, ta, tb in shapes:
    A = rnd(M, K)
    B = rnd(N, K) if tb else rnd(K, N)
    C = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    fn = lambda: ops.gemm(A, B, C, M, N, K, A.shape[1], B.shape[1], N, ta, tb, accumulate=False, split_k=1)
    out = []
    for label, mt, bn in (('default', None, 0), ('256x192', 1, 192), ('256x256', 1, 256)):
        if mt is not None:
            ops.set_option('gemm256_min_tiles', mt)
        ops.set_option('gemm256_bn', bn)
        med, mn = timeit(fn, reps=30)
        out.append('%s %6.1f us %6.1f TF/s' % (label, med, 2.0 * M * N * K / med / 1e6))
        ops.set_option('gemm256_min_tiles', int(os.environ.get('NEURST_GEMM256_MIN_TILES', '128')))
        ops.set_option('gemm256_bn', 0)
    print('%-14s M=%5d N=%5d K=%5d  ' % (name, M, N, K) + ' | '.join(out), flush=True)
