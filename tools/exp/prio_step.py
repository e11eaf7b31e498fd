#!/usr/bin/env python
"""Does a high-priority main stream shorten the step?  (the side streams - CTC branch, weight gradients - stay at normal priority)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neurst_amd
neurst_amd.want_hw_queues()
import torch
from neurst_amd.st.layers.arguments import build_args
from neurst_amd.st.models.TransformerTPE import GlobalStep
from neurst_amd.synthetic import make_batch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)
args = build_args(os.path.join(ROOT, 'configs', 'lut.yaml'), dim_output=10000, compute_dtype='bf16')
model = args.Model(GlobalStep(1000), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
host = make_batch(B=32, Traw=1500, V=10000, L=40, U=40, Lb=40, bert_dim=768, seed=1)
batch = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
def run(n):
    for _ in range(5):
        model.train_step_async(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model.train_step_async(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
hi = torch.cuda.Stream(priority=-1)
for rep in range(3):
    print('default stream %.3f ms' % run(30), flush=True)
    with torch.cuda.stream(hi):
        print('high-priority main stream %.3f ms' % run(30), flush=True)
