#!/usr/bin/env python
"""LN forward variants under rocprofv3 --kernel-trace (experiment driver)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import ops
DEV = torch.device('cuda:0')
rows, C = 16000, 768
x = torch.randn(rows, C, device=DEV).bfloat16()
g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
big = torch.empty(128 * 1024 * 1024, device=DEV)
for it in range(6):
    for rate in (0.0, 0.1):
        y = torch.randn(rows, C, device=DEV).bfloat16()
        big.zero_()
        ops.dropout_add_layernorm_fwd(x, y, g, b, rate, 5)
torch.cuda.synchronize()
