#!/usr/bin/env python
"""How much does the step time depend on WHICH streams of torch's pool the model's second streams are?  Burns k pool
streams before the model creates its own (k = argv[2:]), one process per k.  Usage: python tools/exp/stream_slots.py lut 0 1 2 3"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] != '--child':
    for k in sys.argv[2:]:
        out = subprocess.run([sys.executable, __file__, '--child', sys.argv[1], k], capture_output=True, text=True)
        print(out.stdout.strip().split('\n')[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
    sys.exit(0)
cfgname, k = sys.argv[2], int(sys.argv[3])
sys.path.insert(0, ROOT)
import neurst_amd  # noqa
import torch
from neurst_amd.st.layers.arguments import build_args
from neurst_amd.st.models.TransformerTPE import GlobalStep
from neurst_amd.synthetic import make_batch
burn = [torch.cuda.Stream() for _ in range(k)]
for st in burn:      # first use is what makes HIP give the stream a hardware queue
    with torch.cuda.stream(st):
        torch.zeros(8, device='cuda').add_(1)
torch.cuda.synchronize()
args = build_args(os.path.join(ROOT, 'configs', cfgname + '.yaml'), dim_output=10000, compute_dtype='bf16')
model = args.Model(GlobalStep(1000), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
lut = cfgname == 'lut'
host = make_batch(B=32, Traw=1500, V=10000, L=40 if lut else 80, U=40, Lb=40, bert_dim=768 if lut else None, seed=1)
batch = {kk: torch.from_numpy(v).cuda() for kk, v in host.items()}
for _ in range(4):
    model.train_step_async(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    model.train_step_async(batch)
torch.cuda.synchronize()
print('%s burned=%d GPU_MAX_HW_QUEUES=%s: %.2f ms/step' % (cfgname, k, os.environ.get('GPU_MAX_HW_QUEUES'), (time.perf_counter() - t0) / 20 * 1e3))
