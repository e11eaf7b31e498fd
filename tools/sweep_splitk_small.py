"""split-K sweep of the weight (+ bias) gradient GEMM at decoder / shrunk-encoder sizes (K = rows of the batch is short,
the output has few tiles): microseconds per launch for each split."""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools'))
import torch
from bench_kernels import timeit, DEV
from neurst_amd import ops
rnd = lambda *s: (torch.rand(*s, device=DEV) * 2 - 1).to(torch.bfloat16)
for K in (1024, 1280, 2048, 2560):
    for name, M, N in [('out', 768, 768), ('kv', 1536, 768), ('qkv', 2304, 768), ('ffn1', 3072, 768), ('ffn2', 768, 3072)]:
        A, B = rnd(K, M), rnd(K, N)
        C = torch.zeros(M, N, device=DEV)
        bs = torch.zeros(M, device=DEV)
        res = []
        for sk in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            if K // sk < 64: continue
            med, mn = timeit(lambda: ops.gemm_wgrad_bias(A, B, C, bs, 1, M, N, K, M, N, N, split_k=sk))
            res.append('%d:%.1f' % (sk, med))
        print(name, M, N, K, 'pick=%d' % ops.pick_split_k(M, N, K), ' '.join(res), flush=True)
