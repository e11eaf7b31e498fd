#!/usr/bin/env python
"""Decoding step: host enqueue time against device time (is the loop bound by launches or by the kernels?).  argv: beam"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neurst_amd  # noqa
import torch
from neurst_amd.st.layers.arguments import build_args
from neurst_amd.st.models import graph as G
from neurst_amd.st.models.TransformerTPE import GlobalStep
from neurst_amd.st.models.decoders.transformer_decoder import _CachedDecoderStep
from neurst_amd.synthetic import make_batch
beam = int(sys.argv[1]) if len(sys.argv) > 1 else 1
args = build_args(os.path.join(ROOT, 'configs', 'lut.yaml'), dim_output=10000, compute_dtype='bf16')
args.beam_size, args.max_len = beam, 128
model = args.Model(GlobalStep(0), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=False, args=args)
host = make_batch(B=32, Traw=1500, V=10000, L=40, U=40, Lb=40, bert_dim=None, seed=6, ragged=True)
model.infer(host['feature'], host['len_fea'])
enc, len_enc = model.last['encoded'], model.last['len_encoded']
g = model.graph
with G.use_graph(g), G.variable_scope(model.name), G.variable_scope(model.decoder.name or 'decoder'):
    g.begin_forward(None)
    run = _CachedDecoderStep(model.decoder, enc, len_enc, beam=beam, max_steps=128)
    cur = torch.full((32 * beam, 1), 5, dtype=torch.int32, device='cuda')
    for s in range(8):
        run.step(cur, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(8, 128):
        run.step(cur, s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print('beam %d: host enqueue %.3f ms/step, total %.3f ms/step (%d rows)' % (beam, (t1 - t0) / 120 * 1e3, (t2 - t0) / 120 * 1e3, 32 * beam))
