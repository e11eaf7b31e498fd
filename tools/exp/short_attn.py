"""Times the short-query attention shapes of the LUT step (decoder cross-attention 40 x 500, dh 96; KD attention 40 x 500, dh 192)
with the dq + dkv kernel pair and with the fused kernels forced (mha_fused = 2)."""
import sys, torch
sys.path.insert(0, '.')
from neurst_amd import ops, lib
C = lib.load()
DEV = torch.device('cuda:0')

def timeit(fn, reps=30, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for (B, H, Tq, Tk, dh) in [(32, 8, 40, 500, 96), (32, 4, 40, 500, 192), (32, 8, 40, 40, 96)]:
    d = H * dh
    q = torch.randn(B, Tq, d, device=DEV).to(torch.bfloat16)
    kv = torch.randn(B, Tk, 2 * d, device=DEV).to(torch.bfloat16)
    lens = torch.full((B,), Tk, dtype=torch.int32, device=DEV)
    kf = kv.view(-1)
    for fused in (0, 2):
        C.neurst_set_option(b'mha_fused', fused)
        for rate in (0.0, 0.1):
            fwd = lambda: ops.mha_fwd(q.view(-1), kf, kf[d:], d, 2 * d, 2 * d, B, H, Tq, Tk, dh, ops.MASK_KEYPAD, lens, rate, 123, keep_bits=True)
            o, lse, kb = fwd()
            do = torch.randn_like(o)
            dq = torch.empty_like(q); dkv = torch.empty_like(kv); dk = dkv.view(-1)
            bwd = lambda: ops.mha_bwd(q.view(-1), kf, kf[d:], o, do, lse, dq.view(-1), dk, dk[d:], d, 2 * d, 2 * d, d, 2 * d, 2 * d, B, H, Tq, Tk, dh,
                                      ops.MASK_KEYPAD, lens, rate, 123, keep_bits=kb)
            print('B%d H%d %dx%d dh%d fused=%d rate=%.1f  fwd %7.1f us  bwd %7.1f us' % (B, H, Tq, Tk, dh, fused, rate, timeit(fwd), timeit(bwd)))
C.neurst_set_option(b'mha_fused', 1)
