#!/usr/bin/env python
"""A few launches of the wide attention forward / backward at T = 500, dh = 96, B H = 256 (in-step configuration: stored keep
bits, rate 0.1) for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neurst_amd import ops
DEV = torch.device('cuda:0')
B, H, T, dh = 32, 8, 500, 96
d = H * dh
qkv = torch.randn(B, T, 3 * d, device=DEV).to(torch.bfloat16)
lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
t = qkv.view(-1)
rate = float(os.environ.get('RATE', '0.1'))
for _ in range(4):
    o, lse = ops.mha_fwd(t, t[d:], t[2 * d:], 3 * d, 3 * d, 3 * d, B, H, T, T, dh, ops.MASK_KEYPAD, lens, rate, 123)
torch.cuda.synchronize()
