#!/usr/bin/env python
"""Timing-only ablations of the wide attention forward (mha_fwd_ablate: 1 no exp2, 2 no P.V MFMAs, 4 no Q.K MFMAs, 8 no rescale;
results are wrong by construction), rate 0, T = 500, dh = 96, B H = 256."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neurst_amd import ops
from tools.bench_kernels import timeit, DEV
B, H, T, dh = 32, 8, int(os.environ.get('T', '500')), 96
d = H * dh
qkv = torch.randn(B, T, 3 * d, device=DEV).to(torch.bfloat16)
lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
t = qkv.view(-1)
for abl in (0, 1, 2, 4, 8, 15, 0):
    ops.set_option('mha_fwd_ablate', abl)
    f = lambda: ops.mha_fwd(t, t[d:], t[2 * d:], 3 * d, 3 * d, 3 * d, B, H, T, T, dh, ops.MASK_KEYPAD, lens, 0.0, 123)
    med, mn = timeit(f, reps=30)
    print('ablate %2d: med %6.1f us min %6.1f us' % (abl, med, mn), flush=True)
