#!/usr/bin/env python
"""cProfile of the host side of greedy / beam decoding (where do the microseconds per launch go?).  argv: beam"""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neurst_amd  # noqa
import torch
from neurst_amd.st.layers.arguments import build_args
from neurst_amd.st.models.TransformerTPE import GlobalStep
from neurst_amd.synthetic import make_batch
beam = int(sys.argv[1]) if len(sys.argv) > 1 else 1
args = build_args(os.path.join(ROOT, 'configs', 'lut.yaml'), dim_output=10000, compute_dtype='bf16')
args.beam_size, args.max_len = beam, 64
model = args.Model(GlobalStep(0), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=False, args=args)
host = make_batch(B=32, Traw=1500, V=10000, L=40, U=40, Lb=40, bert_dim=None, seed=6, ragged=True)
model.infer(host['feature'], host['len_fea'])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
model.infer(host['feature'], host['len_fea'])
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print(s.getvalue())
