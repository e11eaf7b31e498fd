#!/usr/bin/env python
"""Yardstick only (never on the product path): which vendor-library kernels torch.mm picks for the step's GEMM shapes.
Run under rocprofv3 --kernel-trace; the Tensile kernel names spell out macro tile, MFMA shape and wave layout."""
import torch
DEV = torch.device('cuda:0')
rnd = lambda *s: (torch.rand(*s, device=DEV) * 2 - 1).to(torch.bfloat16)
shapes = [('qkv', 16000, 2304, 768, 0, 1), ('out', 16000, 768, 768, 0, 1), ('ffn1', 16000, 3072, 768, 0, 1), ('ffn2', 16000, 768, 3072, 0, 1),
          ('dffn2', 16000, 3072, 768, 0, 0), ('dffn1', 16000, 768, 3072, 0, 0), ('dqkv', 16000, 768, 2304, 0, 0), ('wffn1', 3072, 768, 16000, 1, 0)]
for name, M, N, K, ta, tb in shapes:
    A = rnd(K, M) if ta else rnd(M, K)
    B = rnd(N, K) if tb else rnd(K, N)
    C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    for _ in range(3):
        torch.mm(A.t() if ta else A, B.t() if tb else B, out=C)
    torch.cuda.synchronize()
