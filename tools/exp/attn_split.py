#!/usr/bin/env python
"""fused attention forward / backward (T = 500, dh = 96, stored dropout decisions) with the work of a (batch, head) pair on one
workgroup or split over several (neurst_set_option mha_split), at batch 16 and 32"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neurst_amd import ops  # noqa: E402
from tools.bench_kernels import timeit, DEV  # noqa: E402

H, T, dh = 8, 500, 96
d = H * dh
for B in (16, 32):
    qkv = torch.randn(B, T, 3 * d, device=DEV).to(torch.bfloat16)
    lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    t = qkv.view(-1)
    for rate in (0.1, 0.0):
        for split in (0, 1):
            ops.set_option('mha_split', split)
            fwd = lambda: ops.mha_fwd(t, t[d:], t[2 * d:], 3 * d, 3 * d, 3 * d, B, H, T, T, dh, ops.MASK_KEYPAD, lens, rate, 123, keep_bits=rate > 0)
            r = fwd()
            o, lse, kb = (r + (None,))[:3]
            do = torch.randn_like(o)
            dqkv = torch.empty_like(qkv)
            dt = dqkv.view(-1)
            bwd = lambda: ops.mha_bwd(t, t[d:], t[2 * d:], o, do, lse, dt, dt[d:], dt[2 * d:], 3 * d, 3 * d, 3 * d, 3 * d, 3 * d, 3 * d, B, H, T, T, dh,
                                      ops.MASK_KEYPAD, lens, rate, 123, keep_bits=kb)
            mf, _ = timeit(fwd, reps=30)
            mb, _ = timeit(bwd, reps=30)
            print('B=%d rate=%.1f split=%d  fwd %6.1f us  bwd %6.1f us' % (B, rate, split, mf, mb), flush=True)
ops.set_option('mha_split', 1)
