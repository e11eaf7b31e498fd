"""Probe: does the chip get more done with TWO independent half-batch train steps in flight on two streams than with one
full-batch step?  (Two model replicas with their own weights: a GPU-side probe only, not a training mode.)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neurst_amd.st.layers.arguments import build_args
from neurst_amd.st.models.TransformerTPE import GlobalStep
from neurst_amd.synthetic import make_batch


def model_and_batch(B, seed, name):
    args = build_args(os.path.join(ROOT, 'configs', 'lut.yaml'), dim_output=10000, compute_dtype='bf16')
    m = args.Model(GlobalStep(1000), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args, name=name)
    host = make_batch(B=B, Traw=1500, V=10000, L=40, U=40, Lb=40, bert_dim=768, seed=seed)
    return m, {k: torch.from_numpy(v).cuda() for k, v in host.items()}


def run(pairs, steps=12, warm=3):
    streams = [torch.cuda.Stream() for _ in pairs]
    def one():
        for (m, b), s in zip(pairs, streams):
            with torch.cuda.stream(s):
                m.train_step_async(b)
    for _ in range(warm):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


full = [model_and_batch(32, 1, 'transformer')]
t = run(full)
print('one chain  B=32      : %.2f ms per 16000 frames' % (t * 1e3), flush=True)
del full
torch.cuda.empty_cache()
two = [model_and_batch(16, 1, 'transformer'), model_and_batch(16, 2, 'transformer')]
t = run(two)
print('two chains B=16 + 16 : %.2f ms per 16000 frames' % (t * 1e3), flush=True)
one16 = two[:1]
t = run(one16)
print('one chain  B=16      : %.2f ms per  8000 frames' % (t * 1e3), flush=True)
