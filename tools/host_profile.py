#!/usr/bin/env python
"""cProfile of the host side of a train step (where do the microseconds per launch go?)."""
import cProfile, pstats, os, sys, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neurst_amd.st.layers.arguments import build_args
from neurst_amd.st.models.TransformerTPE import GlobalStep
from neurst_amd.synthetic import make_batch
cfgname = sys.argv[1] if len(sys.argv) > 1 else 'costt'
args = build_args(os.path.join(ROOT, 'configs', cfgname + '.yaml'), dim_output=10000, compute_dtype='bf16')
model = args.Model(GlobalStep(1000), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
host = make_batch(B=32, Traw=1500, V=10000, L=40 if cfgname == 'lut' else 80, U=40, Lb=40, bert_dim=768 if cfgname == 'lut' else None, seed=1)
batch = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
for _ in range(3):
    model.train_step_async(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    model.train_step_async(batch)
    torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats(os.environ.get('SORT', 'tottime')).print_stats(int(os.environ.get('TOP', '28')))
print(s.getvalue())
