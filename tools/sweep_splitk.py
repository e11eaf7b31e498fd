import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools'))
import torch
from bench_kernels import timeit, DEV
from neurst_amd import ops
rnd = lambda *s: (torch.rand(*s, device=DEV) * 2 - 1).to(torch.bfloat16)
for name, M, N, K in [('ffn1', 3072, 768, 16000), ('qkv', 2304, 768, 16000), ('out', 768, 768, 16000), ('ctc', 768, 10000, 16000), ('dec ffn', 3072, 768, 1280), ('dec out', 768, 768, 1280), ('kv', 1536, 768, 16000)]:
    A, B = rnd(K, M), rnd(K, N)
    C = torch.zeros(M, N, device=DEV)
    res = []
    for sk in (1, 2, 3, 4, 5, 6, 8, 12, 16):
        if K // sk < 128: continue
        med, mn = timeit(lambda: ops.gemm(A, B, C, M, N, K, M, N, N, 1, 0, accumulate=True, split_k=sk))
        res.append('%d:%.0f' % (sk, med))
    print(name, M, N, K, ' '.join(res))
