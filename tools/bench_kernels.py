#!/usr/bin/env python
"""Kernel-level timing (HIP events, interleaved rounds in one process) of the GEMM / attention / LN shapes
of the LUT step.  Usage: python tools/bench_kernels.py [gemm|attn|ln|conv|adam|all]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import ops  # noqa: E402

DEV = torch.device('cuda:0')


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


def bench_gemm():
    rnd = lambda *s: (torch.rand(*s, device=DEV) * 2 - 1).to(torch.bfloat16)
    shapes = [  # name, M, N, K, ta, tb, c fp32?, split
        ('fwd qkv   NT', 16000, 2304, 768, 0, 1, False, 1),
        ('fwd out   NT', 16000, 768, 768, 0, 1, False, 1),
        ('fwd ffn1  NT', 16000, 3072, 768, 0, 1, False, 1),
        ('fwd ffn2  NT', 16000, 768, 3072, 0, 1, False, 1),
        ('ctc head  NN', 16000, 10000, 768, 0, 0, True, 1),
        ('dgrad ffn2 NN', 16000, 3072, 768, 0, 0, False, 1),
        ('dgrad ffn1 NN', 16000, 768, 3072, 0, 0, False, 1),
        ('dgrad qkv NN', 16000, 768, 2304, 0, 0, False, 1),
        ('dgrad ctc NT', 16000, 768, 10000, 0, 1, False, 1),
        ('wgrad ffn1 TN', 3072, 768, 16000, 1, 0, True, None),
        ('wgrad ffn2 TN', 768, 3072, 16000, 1, 0, True, None),
        ('wgrad qkv TN', 2304, 768, 16000, 1, 0, True, None),
        ('wgrad out TN', 768, 768, 16000, 1, 0, True, None),
        ('wgrad ctc TN', 768, 10000, 16000, 1, 0, True, None),
        ('dec small NT', 1280, 768, 768, 0, 1, False, 1),
        ('dec kv    NT', 16000, 1536, 768, 0, 1, False, 1),
    ]
    for name, M, N, K, ta, tb, cf32, split in shapes:
        A = rnd(K, M) if ta else rnd(M, K)
        B = rnd(N, K) if tb else rnd(K, N)
        C = torch.zeros(M, N, device=DEV, dtype=torch.float32 if cf32 else torch.bfloat16)
        sk = ops.pick_split_k(M, N, K) if split is None else split
        fn = lambda: ops.gemm(A, B, C, M, N, K, A.shape[1], B.shape[1], N, ta, tb, accumulate=sk > 1 or split is None, split_k=sk)
        med, mn = timeit(fn)
        fl = 2.0 * M * N * K
        line = '%-16s M=%5d N=%5d K=%5d split=%2d  med %8.1f us  min %8.1f us  %7.1f TF/s' % (name, M, N, K, sk, med, mn, fl / med / 1e6)
        if os.environ.get('YARDSTICK'):   # the vendor library on the same shape: a yardstick for this tool only, never on the product path
            At, Bt = (A.t() if ta else A), (B.t() if tb else B)
            Cl = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            lm, _ = timeit(lambda: torch.mm(At, Bt, out=Cl))
            line += '   | library %8.1f us %7.1f TF/s' % (lm, fl / lm / 1e6)
        print(line)


def bench_wgrad():
    """weight-gradient shapes: plain TN split-K against the same launch with the fused bias gradient, per split"""
    rnd = lambda *s: (torch.rand(*s, device=DEV) * 2 - 1).to(torch.bfloat16)
    shapes = [('ffn1', 3072, 768, 16000, 1), ('ffn2', 768, 3072, 16000, 1), ('qkv', 2304, 768, 16000, 1), ('out', 768, 768, 16000, 1)]
    if os.environ.get('DECODER_SHAPES'):      # weight gradients of the decoder blocks: K = B * L = 1280 rows
        shapes = [('dec out', 768, 768, 1280, 1), ('dec qkv', 2304, 768, 1280, 1), ('dec ffn1', 3072, 768, 1280, 1), ('dec ffn2', 768, 3072, 1280, 1)]
    for name, M, N, K, which in shapes:
        A, B = rnd(K, M), rnd(K, N)          # A = dY [rows, out], B = x [rows, in]
        C = torch.zeros(M, N, device=DEV, dtype=torch.float32)
        bsum = torch.zeros(M, device=DEV, dtype=torch.float32)
        fl = 2.0 * M * N * K
        cs, _ = timeit(lambda: ops.colsum(A, bsum))
        for sk in (1, 2, 3, 4, 5, 6, 8):
            plain, _ = timeit(lambda: ops.gemm(A, B, C, M, N, K, M, N, N, 1, 0, accumulate=True, split_k=sk))
            fused, _ = timeit(lambda: ops.gemm_wgrad_bias(A, B, C, bsum, which, M, N, K, M, N, N, split_k=sk))
            print('wgrad %-5s M=%5d N=%5d K=%5d split=%d%s  plain %7.1f us %6.1f TF/s | +bias-grad %7.1f us %6.1f TF/s | colsum alone %5.1f us' % (
                name, M, N, K, sk, '*' if sk == ops.pick_split_k(M, N, K) else ' ', plain, fl / plain / 1e6, fused, fl / fused / 1e6, cs), flush=True)


def bench_attn():
    B, H, T, dh = 32, 8, 500, 96
    d = H * dh
    qkv = (torch.randn(B, T, 3 * d, device=DEV)).to(torch.bfloat16)
    lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    t = qkv.view(-1)
    for rate in (0.0, 0.1):
        fwd = lambda: ops.mha_fwd(t, t[d:], t[2 * d:], 3 * d, 3 * d, 3 * d, B, H, T, T, dh, ops.MASK_KEYPAD, lens, rate, 123)
        o, lse = fwd()
        do = torch.randn_like(o)
        dqkv = torch.empty_like(qkv)
        dt = dqkv.view(-1)
        bwd = lambda: ops.mha_bwd(t, t[d:], t[2 * d:], o, do, lse, dt, dt[d:], dt[2 * d:], 3 * d, 3 * d, 3 * d, 3 * d, 3 * d, 3 * d, B, H, T, T, dh,
                                  ops.MASK_KEYPAD, lens, rate, 123)
        fl = 4.0 * B * H * T * T * dh
        m, _ = timeit(fwd)
        print('attn fwd rate=%.1f  %8.1f us  %6.1f TF/s' % (rate, m, fl / m / 1e6))
        m, _ = timeit(bwd)
        print('attn bwd rate=%.1f  %8.1f us  %6.1f TF/s (10 B H T^2 dh flops)' % (rate, m, 2.5 * fl / m / 1e6))


def bench_ln():
    rows, C = int(os.environ.get('LN_ROWS', '16000')), 768
    x = torch.randn(rows, C, device=DEV).to(torch.bfloat16)
    y = torch.randn(rows, C, device=DEV).to(torch.bfloat16)
    g = torch.ones(C, device=DEV)
    b = torch.zeros(C, device=DEV)
    from neurst_amd import lib
    pfs = [int(v) for v in os.environ.get('LN_PF', '0,3').split(',')]
    for pf, rate in [(p_, r_) for r_ in (0.0, 0.1, 0.1) for p_ in pfs]:
        lib.load().neurst_set_option(b'ln_prefetch', pf)
        print('ln_prefetch', pf, end=': ')
        f = lambda: ops.dropout_add_layernorm_fwd(x, y, g, b, rate, 5)
        m, _ = timeit(f)
        print('ln fwd rate=%.1f %7.1f us  %6.0f GB/s (4*rows*C*2 B)' % (rate, m, 4 * rows * C * 2 / m / 1e3))
        out, mean, rstd = f()
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        f2 = lambda: ops.dropout_add_layernorm_bwd(out, y, g, mean, rstd, dg, db, rate, 5)
        m, _ = timeit(f2)
        print('ln bwd rate=%.1f %7.1f us  %6.0f GB/s (%d*rows*C*2 B)' % (rate, m, (4 if rate else 3) * rows * C * 2 / m / 1e3, 4 if rate else 3))


def bench_adam():
    """the LUT model's arenas: 120.3 M fp32 parameters, 30 B per parameter and step (p, g, m, v in; p, m, v, bf16 copy out)"""
    n = int(os.environ.get('ADAM_N', '120300000'))
    p, g = torch.randn(n, device=DEV), torch.randn(n, device=DEV) * 1e-3
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    sh = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    med, mn = timeit(lambda: ops.adam_step(p, g, m, v, sh, 1e-4, 0.9, 0.98, 1e-9, 1.0), reps=15)
    print('adam n=%d  med %7.1f us  min %7.1f us  %6.0f GB/s (30 B per parameter)' % (n, med, mn, 30.0 * n / med / 1e3))


def bench_conv():
    """a4: the two conv + LN + ReLU layers of a `use_cnn_resample` front end at BASELINE config 2's input (32 x 1500 x 80),
    3 x 3 kernels, stride 2, 64 filters (transformer_encoder.py:69-95), and the 3 x 3 max pool behind them.  Algorithmic
    bytes: forward reads x and writes conv_out + out; backward reads dout, conv_out, out, x and writes dx."""
    B = int(os.environ.get('CONV_B', '32'))
    for name, H, W, Cin, Cout, need_dx in (('conv_0', 1500, 80, int(os.environ.get('CONV_CIN', '1')), 64, False), ('conv_1', 750, 40, 64, 64, True)):
        x = torch.randn(B, H, W, Cin, device=DEV).to(torch.bfloat16)
        w = (torch.randn(3, 3, Cin, Cout, device=DEV) * 0.1)
        bias, gam, bet = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
        f = lambda: ops.conv2d_ln_relu_fwd(x, w, bias, gam, bet, (3, 3), (2, 2))
        out, conv_out, mean, rstd = f()
        m, _ = timeit(f, reps=10)
        nb = (x.numel() + 2 * out.numel()) * 2
        fl = 2.0 * out.numel() * 9 * Cin
        print('%s fwd  [%d,%d,%d,%d]->%d  %8.1f us  %6.0f GB/s  %6.1f TF/s' % (name, B, H, W, Cin, Cout, m, nb / m / 1e3, fl / m / 1e6))
        dout = torch.randn_like(out)
        dw, db, dg, dbe = torch.zeros_like(w), torch.zeros(Cout, device=DEV), torch.zeros(Cout, device=DEV), torch.zeros(Cout, device=DEV)
        f2 = lambda: ops.conv2d_ln_relu_bwd(dout, x, w, gam, conv_out, out, mean, rstd, dw, db, dg, dbe, (3, 3), (2, 2), need_dx)
        m, _ = timeit(f2, reps=5)
        nb = (3 * out.numel() + x.numel() * (2 if need_dx else 1)) * 2
        print('%s bwd  (dx %s)             %8.1f us  %6.0f GB/s  %6.1f TF/s' % (name, need_dx, m, nb / m / 1e3, fl * (2 if need_dx else 1) / m / 1e6))
    y = torch.randn(B, 375, 20, 64, device=DEV).to(torch.bfloat16)
    f3 = lambda: ops.maxpool3x3_fwd(y)
    m, _ = timeit(f3)
    print('maxpool fwd [%d,375,20,64]       %8.1f us  %6.0f GB/s' % (B, m, y.numel() * 5 / m / 1e3))
    o, arg = f3()
    m, _ = timeit(lambda: ops.maxpool3x3_bwd(o, arg))
    print('maxpool bwd                      %8.1f us  %6.0f GB/s' % (m, y.numel() * 5 / m / 1e3))


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what == 'wgrad':
        bench_wgrad()
    if what in ('gemm', 'all'):
        bench_gemm()
    if what in ('attn', 'all'):
        bench_attn()
    if what in ('ln', 'all'):
        bench_ln()
    if what in ('adam', 'all'):
        bench_adam()
    if what in ('conv', 'all'):
        bench_conv()
