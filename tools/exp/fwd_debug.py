import sys, numpy as np, torch
sys.path.insert(0, '.')
from neurst_amd import ops
from oracle import ops as O
DEV = torch.device('cuda:0')
B, H, T, dh = 2, 8, 130, 96
d = H * dh
rng = np.random.default_rng(0)
q = (rng.standard_normal((B, T, d)) * 0.5).astype(np.float32)
k = (rng.standard_normal((B, T, d)) * 0.5).astype(np.float32)
v = (rng.standard_normal((B, T, d)) * 0.5).astype(np.float32)
bf = lambda a: torch.from_numpy(a).to(torch.bfloat16)
qd, kd, vd = bf(q).to(DEV), bf(k).to(DEV), bf(v).to(DEV)
q, k, v = bf(q).float().numpy(), bf(k).float().numpy(), bf(v).float().numpy()
lens = np.array([130, 77], np.int32)
ld = torch.from_numpy(lens).to(DEV)
for rate in (0.0, 0.1):
    ref, cache = O.mha_core_fwd(q.astype(np.float64), k.astype(np.float64), v.astype(np.float64), H, O.MASK_KEYPAD, lens, rate, 77)
    o, lse, kb = ops.mha_fwd(qd, kd, vd, d, d, d, B, H, T, T, dh, ops.MASK_KEYPAD, ld, rate, 77, keep_bits=True)
    err = np.abs(o.float().cpu().numpy() - ref)
    print('rate', rate, 'max err', err.max(), 'ref max', np.abs(ref).max())
    if rate > 0:
        nw = (T + 31) // 32
        want = O.attn_keep_words(77, B * H * T, nw, rate)
        got = kb.cpu().numpy().view(np.uint32).reshape(B * H * T, nw)
        print('bitmap equal rows:', (got == want).all(1).mean(), 'per word col:', (got == want).mean(0))
        e = err.reshape(B, T, H, dh).max(-1)
        print('err by (b, h):', e.max(1).round(3))
        print('err by q (b=0):', e[0].max(1).round(2)[:40])
