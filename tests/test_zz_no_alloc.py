"""The allocation-free entry point neurst_dropout_add_layernorm_bwd_noalloc (include/neurst_hip.h) against the existing
neurst_dropout_add_layernorm_bwd with a caller workspace: same kernels, the block partials in the library's built-in
per-GPU array instead.  Bit-exact (both reductions are deterministic), on two calls in a row, the second on a smaller
input, so stale block partials of the first call would show.  Collected last on purpose (file name)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(C_lib, rows, C, rate, dtype, noalloc, seed):
    from neurst_amd import lib
    g = torch.Generator(device='cpu').manual_seed(seed)
    td = torch.bfloat16 if dtype == lib.BF16 else torch.float32
    dev = 'cuda:0'
    dout = torch.randn(rows, C, generator=g).to(td).to(dev)
    z = torch.randn(rows, C, generator=g).to(td).to(dev)
    gamma = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev)
    mean = z.float().mean(-1).contiguous()
    rstd = (1.0 / torch.sqrt(z.float().var(-1, unbiased=False) + 1e-6)).contiguous()
    dz = torch.full_like(z, float('nan'))
    dy = torch.full_like(z, float('nan'))
    dgamma = torch.full((C,), 0.25, device=dev)
    dbeta = torch.full((C,), -0.5, device=dev)
    dbias = torch.zeros(C, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    if noalloc:
        rc = C_lib.neurst_dropout_add_layernorm_bwd_noalloc(p(dout), p(z), p(gamma), p(mean), p(rstd), p(dz), p(dy), p(dgamma), p(dbeta),
                                                            p(dbias), rows, C, rate, 1234, None, 0, dtype, stream)
    else:
        nblk = C_lib.neurst_layernorm_bwd_blocks(rows)
        partial = torch.full((2 * nblk * C,), float('nan'), device=dev)
        rc = C_lib.neurst_dropout_add_layernorm_bwd(p(dout), p(z), p(gamma), p(mean), p(rstd), p(dz), p(dy), p(dgamma), p(dbeta),
                                                    p(partial), p(dbias), rows, C, rate, 1234, None, 0, dtype, stream)
    assert rc == 0, C_lib.neurst_last_error()
    torch.cuda.synchronize()
    return [t.float().cpu().numpy() for t in (dz, dy, dgamma, dbeta, dbias)]


def test_no_alloc_layernorm_bwd_matches_workspace_entry_point_twice():
    from neurst_amd import lib
    C_lib = lib.load()
    # first call: BASELINE config 2's residual shape; second call: fewer rows and columns (fewer block slots, all re-written)
    for rows, C, rate, dtype, seed in ((16000, 768, 0.1, lib.BF16, 1), (1280, 512, 0.1, lib.BF16, 2), (77, 64, 0.0, lib.F32, 3)):
        want = _run(C_lib, rows, C, rate, dtype, False, seed)
        alloc_before = torch.cuda.memory_allocated()
        got = _run(C_lib, rows, C, rate, dtype, True, seed)
        assert torch.cuda.memory_allocated() == alloc_before      # (torch's own view: the test's tensors were freed again)
        names = ('dz', 'dy', 'dgamma', 'dbeta', 'dbias')
        for n, a, b in zip(names, want, got):
            if n == 'dy' and rate == 0.0:
                continue                                           # not written without dropout
            if n == 'dbias':                                       # summed with fp32 atomics in both entry points
                np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-3, err_msg=f'{n} rows={rows}')
            else:
                assert np.array_equal(a, b), f'{n} differs at rows={rows} C={C}'
        assert np.isfinite(got[2]).all() and np.isfinite(got[3]).all()


def test_no_alloc_rejects_sizes_beyond_its_scratch():
    from neurst_amd import lib
    C_lib = lib.load()
    rows, C = 16000, 4096      # 2 * 256 blocks * 4096 columns = 2 Mi floats > 1 Mi
    assert 2 * C_lib.neurst_layernorm_bwd_blocks(rows) * C > (1 << 20)
    t = torch.zeros(8, device='cuda:0')
    p = ctypes.c_void_p(t.data_ptr())
    rc = C_lib.neurst_dropout_add_layernorm_bwd_noalloc(p, p, p, p, p, p, p, p, p, None, rows, C, 0.0, 1, None, 0, lib.F32, None)
    assert rc == -1 and b'scratch' in C_lib.neurst_last_error()
