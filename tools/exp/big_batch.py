#!/usr/bin/env python
"""Maximum sizes (BASELINE config 5: long utterances, batch sized for 288 GB): a batch of 32 x 3000 raw frames repeated 8
times (B = 256, T = 1000: 2.56e9 CTC logits - past 32-bit indexing) must give the losses and the gradient of the batch
of 32 (means over identical utterances), dropout off.  Prints peak memory.  argv: repeat factor (default 8), raw frames (3000), base batch (32)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neurst_amd  # noqa
import numpy as np
import torch
from neurst_amd.st.layers.arguments import build_args, load_config
from neurst_amd.st.models.TransformerTPE import GlobalStep
from neurst_amd.synthetic import make_batch
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 8
TRAW = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
B0 = int(sys.argv[3]) if len(sys.argv) > 3 else 32
cfg = load_config(os.path.join(ROOT, 'configs', 'lut.yaml'))
for part in ('encoder', 'encoder2', 'decoder'):
    cfg['model'][part]['attention_dropout_rate'] = 0.0
    cfg['model'][part]['residual_dropout_rate'] = 0.0
cfg['model']['kd']['attention_dropout_rate'] = 0.0
args = build_args(cfg, dim_output=10000, compute_dtype='bf16')
model = args.Model(GlobalStep(1000), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
host = make_batch(B=B0, Traw=TRAW, V=10000, L=40, U=40, Lb=40, bert_dim=768, seed=11, ragged=True)
small = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
a = model.forward_backward(small).clone()
ga = model.store.grads.clone()
big = {k: v.repeat((rep,) + (1,) * (v.dim() - 1)).contiguous() for k, v in small.items()}
torch.cuda.synchronize()
t0 = time.perf_counter()
b = model.forward_backward(big).clone()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
gb = model.store.grads
rel = (gb - ga).norm().item() / ga.norm().item()
cos = torch.dot(gb, ga).item() / (gb.norm().item() * ga.norm().item())
print('Traw=%d B=%d: losses %s | B=%d: %s' % (TRAW, B0 * rep, np.array2string(b.cpu().numpy(), precision=4), B0, np.array2string(a.cpu().numpy(), precision=4)))
print('gradient: relative L2 difference %.2e, cosine %.6f; step %.1f ms; peak memory %.1f GB' % (rel, cos, dt * 1e3, torch.cuda.max_memory_allocated() / 2**30))
assert torch.isfinite(b).all() and torch.allclose(a, b, rtol=2e-3), 'losses differ'
assert rel < 2e-2 and cos > 0.9995, 'gradients differ'
print('big batch ok')
