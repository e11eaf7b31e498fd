#!/usr/bin/env python
"""Run under `rocprofv3 --kernel-trace`: launches the LN / attention kernels of the LUT step a few times each so
their device durations can be read from the trace (events around single short launches are too noisy)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import ops  # noqa: E402

DEV = torch.device('cuda:0')
rows, C = 16000, 768
x = torch.randn(rows, C, device=DEV).bfloat16()
g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
big = torch.empty(256 * 1024 * 1024, device=DEV)      # flush the infinity cache between launches
for it in range(6):
    for rate in (0.0, 0.1):
        y = torch.randn(rows, C, device=DEV).bfloat16()
        big.zero_()
        out, mean, rstd = ops.dropout_add_layernorm_fwd(x, y, g, b, rate, 5)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        big.zero_()
        ops.dropout_add_layernorm_bwd(out, y, g, mean, rstd, dg, db, rate, 5)
B, H, T, dh = 32, 8, 500, 96
d = H * dh
qkv = torch.randn(B, T, 3 * d, device=DEV).bfloat16()
lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
t = qkv.view(-1)
for it in range(4):
    big.zero_()
    o, lse = ops.mha_fwd(t, t[d:], t[2 * d:], 3 * d, 3 * d, 3 * d, B, H, T, T, dh, ops.MASK_KEYPAD, lens, 0.1, 123)
    do = torch.randn_like(o)
    dqkv = torch.empty_like(qkv)
    dt = dqkv.view(-1)
    big.zero_()
    ops.mha_bwd(t, t[d:], t[2 * d:], o, do, lse, dt, dt[d:], dt[2 * d:], 3 * d, 3 * d, 3 * d, 3 * d, 3 * d, 3 * d, B, H, T, T, dh,
                ops.MASK_KEYPAD, lens, 0.1, 123)
torch.cuda.synchronize()
