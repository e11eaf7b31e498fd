"""-m gpu: seeded random walk over the GEMM dispatcher (csrc/gemm.hip launch_gemm, csrc/gemm256.hip) through the C ABI.

The per-kernel tests of test_gpu_ops.py fix one flag combination per shape; here every case draws its shape from the
size class of ONE dispatch bucket (register-staged, thin 32- / 64-row tiles, 4-stage small grids, 2-stage mid-size grids with
their compile-time epilogue variants, persistent, 256-row tiles) and then draws bias / relu / relu-gradient gate / accumulate /
output type / padded leading dimensions independently, so the combinations the step never uses (gate AND accumulate, padded
ldc on the straight-line epilogues, ...) are compared with fp64 numpy as well.  `dense`, /root/reference/st/models/tools/utils.py:367-398.
Tolerances: fp32 output of bf16 operands 1e-5 of the result's maximum (fp32 accumulation against fp64), bf16 output 1e-2
(one bf16 rounding of the result, 2^-8, on top)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
FUZZ_SCALE = int(os.environ.get('NEURST_FUZZ_SCALE', '1'))     # more seeds for a longer hunt (not the committed suite)

if torch.cuda.is_available():
    from neurst_amd import ops
    DEV = torch.device('cuda:0')


def _bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).float().numpy()


def _draw_shape(rng, bucket, seed):
    """(M, N, K, trans_a, trans_b, aligned) of one dispatch bucket; tiles are 128 x 128 unless stated"""
    r = lambda lo, hi: int(rng.integers(lo, hi + 1))
    if bucket == 'staged':            # nothing aligned: register-staged kernels, scalar stores
        return r(1, 300), r(1, 300), r(1, 400), seed & 1, (seed >> 1) & 1, False   # the four layouts in turn
    if bucket == 'thin32':            # <= 64 tiles, K-contiguous A
        tm, tn = r(1, 8), r(1, 8)
        return 128 * tm - r(0, 127), 128 * tn - 8 * r(0, 15), 8 * r(32, 160), 0, r(0, 1), True
    if bucket == 'thin64':            # 65 .. 128 tiles
        tm = r(9, 16)
        tn = r(-(-65 // tm), 128 // tm)
        return 128 * tm - r(0, 127), 128 * tn - 8 * r(0, 15), 8 * r(32, 96), 0, r(0, 1), True
    if bucket == 'small4':            # <= 256 tiles with k-strided A (weight-gradient layout), 4-stage pipeline
        tm, tn = r(1, 12), r(1, 12)
        return 128 * tm - 8 * r(0, 15), 128 * tn - 8 * r(0, 15), 8 * r(32, 200), 1, 0, True
    if bucket == 'mid':               # 257 .. 1023 tiles: 2-stage kernels and their epilogue variants
        tm = r(17, 60)
        tn = r(-(-257 // tm), 1023 // tm)
        return 128 * tm - (8 * r(0, 15) if seed % 4 == 3 else r(0, 127)), 128 * tn - 8 * r(0, 15), 8 * r(8, 40), int(seed % 4 == 3), int(seed % 4 in (0, 2)), True   # NT, NN, NT, TN in turn
    if bucket == 'persist':           # >= 1024 tiles, NT, whole 64-column K-tiles
        tm = r(32, 48)
        tn = r(-(-1024 // tm), 1400 // tm)
        return 128 * tm - r(0, 127), 128 * tn - 8 * r(0, 15), 64 * r(3, 6), 0, 1, True
    if bucket == 'long256':           # >= 128 tiles of 256 x 256, NT, K >= 2560
        return 256 * 16 - r(0, 255), 256 * 8 - 8 * r(0, 31), 8 * r(320, 360), 0, 1, True
    if bucket == 'wide256':           # fp32 result at least 4096 wide, NN
        return 256 * 8 - r(0, 255), 256 * 16 + 8 * r(0, 40), 8 * r(8, 48), 0, 0, True
    raise AssertionError(bucket)


BUCKETS = ['staged', 'thin32', 'thin64', 'small4', 'mid', 'persist', 'long256', 'wide256']


@pytest.mark.parametrize('bucket', BUCKETS)
@pytest.mark.parametrize('seed', range(6 * FUZZ_SCALE))
def test_gemm_dispatch_random_flags(bucket, seed):
    rng = np.random.default_rng(7919 * seed + BUCKETS.index(bucket))
    M, N, K, ta, tb, aligned = _draw_shape(rng, bucket, seed)
    pad = (lambda: 8 * int(rng.integers(0, 3))) if aligned else (lambda: int(rng.integers(0, 4)))
    c_f32 = bucket == 'wide256' or bool(rng.integers(0, 2))
    use_bias, relu, gate, accumulate = (bool(rng.integers(0, 2)) for _ in range(4))
    a_rows, a_cols = (K, M) if ta else (M, K)
    b_rows, b_cols = (N, K) if tb else (K, N)
    lda, ldb, ldc = a_cols + pad(), b_cols + pad(), N + pad()
    A = _bf16(rng.standard_normal((a_rows, a_cols)))
    B = _bf16(rng.standard_normal((b_rows, b_cols)))
    bias = rng.standard_normal(N).astype(np.float32)
    G = _bf16(rng.standard_normal((M, N)))
    C0 = _bf16(rng.standard_normal((M, N)))
    bf = torch.bfloat16
    tc = torch.float32 if c_f32 else bf

    def padded(a, ld, dtype, fill):
        t = torch.full((a.shape[0], ld), fill, dtype=dtype, device=DEV)
        t[:, :a.shape[1]] = torch.from_numpy(a).to(DEV).to(dtype)
        return t

    dA, dB = padded(A, lda, bf, 7.0), padded(B, ldb, bf, -7.0)       # the padding must never be read into the product
    dG = padded(G, ldc, bf, 1.0)
    dC = padded(C0, ldc, tc, 123.0)
    ops.gemm(dA, dB, dC, M, N, K, lda, ldb, ldc, ta, tb, bias=torch.from_numpy(bias).to(DEV) if use_bias else None, relu=relu,
             relu_src=dG if gate else None, accumulate=accumulate)
    torch.cuda.synchronize()
    ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
    if use_bias:
        ref = ref + bias
    if relu:
        ref = np.maximum(ref, 0)
    if gate:
        ref = ref * (G > 0)
    if accumulate:
        ref = ref + C0
    got = dC.float().cpu().numpy()
    what = dict(bucket=bucket, M=M, N=N, K=K, ta=ta, tb=tb, lda=lda, ldb=ldb, ldc=ldc, c_f32=c_f32, bias=use_bias, relu=relu, gate=gate,
                accumulate=accumulate, label=ops.gemm_kernel_label(dA, dC, M, N, K, ta, tb))
    err = np.abs(got[:, :N] - ref).max() / max(np.abs(ref).max(), 1e-12)
    assert err < (1e-5 if c_f32 else 1e-2), (err, what)
    assert np.all(got[:, N:] == 123.0), ('columns beyond N were written', what)
