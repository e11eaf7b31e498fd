#!/usr/bin/env python
"""How far ahead of the GPU does the host run?  Enqueue time per step (no sync) vs device time per step."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neurst_amd.st.layers.arguments import build_args
from neurst_amd.st.models.TransformerTPE import GlobalStep
from neurst_amd.synthetic import make_batch
cfgname = sys.argv[1] if len(sys.argv) > 1 else 'lut'
args = build_args(os.path.join(ROOT, 'configs', cfgname + '.yaml'), dim_output=10000, compute_dtype='bf16')
model = args.Model(GlobalStep(1000), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
BATCH = int(os.environ.get('BATCH', '32'))
host = make_batch(B=BATCH, Traw=1500, V=10000, L=40 if cfgname == 'lut' else 80, U=40, Lb=40, bert_dim=768 if cfgname == 'lut' else None, seed=1)
batch = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
for leg in (3, 6):      # COSTT: the shrink bound settles after the first finished steps
    for _ in range(leg):
        model.train_step_async(batch)
    torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    model.train_step_async(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(cfgname, 'B =', BATCH, 'host enqueue %.2f ms/step, total %.2f ms/step' % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
# one step enqueued on an idle GPU: pure host cost
torch.cuda.synchronize()
t0 = time.perf_counter(); model.train_step_async(batch); t1 = time.perf_counter(); torch.cuda.synchronize()
print('single step enqueue on idle GPU %.2f ms' % ((t1 - t0) * 1e3))
