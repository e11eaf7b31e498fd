#!/usr/bin/env python
"""Does a HIP stream created with a CU mask confine kernels to part of the chip on this box?  Times one large GEMM
(ops.gemm, 16000 x 3072 x 768) on an unmasked stream and on streams masked to 192 / 128 / 64 CUs."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neurst_amd import ops

hip = ctypes.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(ncu, total=256):
    words = (total + 31) // 32
    bits = (1 << ncu) - 1
    arr = (ctypes.c_uint32 * words)(*[(bits >> (32 * i)) & 0xffffffff for i in range(words)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


torch.cuda.init()
x = torch.randn(16000, 768, device='cuda').bfloat16()
w = torch.randn(3072, 768, device='cuda').bfloat16()
out = torch.empty(16000, 3072, device='cuda', dtype=torch.bfloat16)


def run(stream, label):
    with torch.cuda.stream(stream):
        for _ in range(3):
            ops.gemm(x, w, out, 16000, 3072, 768, 768, 768, 3072, False, True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            ops.gemm(x, w, out, 16000, 3072, 768, 768, 768, 3072, False, True)
        e1.record(stream)
    torch.cuda.synchronize()
    print('%-12s %.1f us per GEMM' % (label, e0.elapsed_time(e1) * 100), flush=True)


run(torch.cuda.Stream(), 'unmasked')
for n in (192, 128, 64):
    run(masked_stream(n), 'mask %d' % n)

# ---- does a masked heavy stream leave the rest of the chip to a chain of small dependent launches on another stream?
xs = torch.randn(1280, 768, device='cuda').bfloat16()
ws = torch.randn(768, 768, device='cuda').bfloat16()
os_ = torch.empty(1280, 768, device='cuda', dtype=torch.bfloat16)


def light(n=200):
    for _ in range(n):
        ops.gemm(xs, ws, os_, 1280, 768, 768, 768, 768, 768, False, True)


def heavy(n=10):
    for _ in range(n):
        ops.gemm(x, w, out, 16000, 3072, 768, 768, 768, 3072, False, True)


def both(hs, label):
    main = torch.cuda.current_stream()
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if hs is not None:
            with torch.cuda.stream(hs):
                heavy()
        light()
        if hs is not None:
            main.wait_stream(hs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
    print('%-40s %.2f ms' % (label, dt), flush=True)


torch.cuda.synchronize(); t0 = time.perf_counter(); light(); torch.cuda.synchronize(); t0 = time.perf_counter(); light(); torch.cuda.synchronize()
print('light chain alone                        %.2f ms' % ((time.perf_counter() - t0) * 1e3))
both(None, 'light chain alone (again)')
both(torch.cuda.Stream(), 'light + heavy on an unmasked stream')
for n in (192, 128, 64):
    both(masked_stream(n), 'light + heavy on a %d-CU stream' % n)
