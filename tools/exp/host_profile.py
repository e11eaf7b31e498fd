"""Where the host spends a step's enqueue time: cProfile over a few train steps (no synchronisation inside the profiled
region, so the figures are the Python + ctypes + launch cost, not GPU time).
Usage: python tools/exp/host_profile.py configs/costt.yaml [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neurst_amd  # noqa: E402

neurst_amd.want_hw_queues()
from neurst_amd.st.layers.arguments import build_args  # noqa: E402
from neurst_amd.st.models.TransformerTPE import GlobalStep  # noqa: E402
from neurst_amd.synthetic import make_batch  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'configs', 'costt.yaml')
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
args = build_args(cfg, dim_output=10000, compute_dtype='bf16')
is_lut = args.model.loss_type == 'ctc_kd_ce'
model = args.Model(GlobalStep(1000), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
host = make_batch(B=32, Traw=1500, V=10000, L=40 if is_lut else 80, U=40, Lb=40, bert_dim=768 if is_lut else None, seed=1234)
batch = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
for _ in range(3):
    model.train_step_async(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    model.train_step_async(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.2f ms/step, with the GPU drained %.2f ms/step' % ((t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    model.train_step_async(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(int(os.environ.get('TOP', '45')))
if os.environ.get('CUM'):
    st.sort_stats('cumtime').print_stats(int(os.environ.get('TOP', '45')))
