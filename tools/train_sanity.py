#!/usr/bin/env python
"""Optimisation sanity at full width: the LUT (or COSTT) train step in bf16 repeated on ONE synthetic batch (B utterances) -
the total loss and each of its parts must fall steadily and stay finite (the model memorises the batch).  Not a
benchmark; evidence that forward, backward, the loss mix, the lr schedule and Adam work together at d=768 / V=10000.
Usage: python tools/train_sanity.py [lut|costt] [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neurst_amd.st.layers.arguments import build_args  # noqa: E402
from neurst_amd.st.models.TransformerTPE import GlobalStep  # noqa: E402
from neurst_amd.synthetic import make_batch  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'lut'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    args = build_args(os.path.join(ROOT, 'configs', name + '.yaml'), dim_output=10000, compute_dtype='bf16')
    args.warmup_steps, args.peak, args.decay_steps = 40, 4e-4, 100000
    lut = args.model.loss_type == 'ctc_kd_ce'
    model = args.Model(GlobalStep(0), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
    host = make_batch(B=16, Traw=600, V=10000, L=20 if lut else 40, U=20, Lb=20, bert_dim=768 if lut else None, seed=3, ragged=True)
    batch = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
    hist = []
    t0 = time.perf_counter()
    for s in range(steps):
        loss, _, l_ast, l_ctc, l_kd = model.train_step(batch)
        hist.append((loss, l_ast, l_ctc, l_kd))
        if s % 25 == 0 or s == steps - 1:
            print('step %4d  loss %10.3f  ast %8.3f  ctc %10.3f  kd %8.4f  lr %.2e' % (s, loss, l_ast, l_ctc, l_kd, model.last['lr']), flush=True)
    h = np.array(hist)
    assert np.isfinite(h).all(), 'non-finite loss'
    first, last = h[:10].mean(0), h[-10:].mean(0)
    print('mean of first / last 10 steps: loss %.2f -> %.2f, ast %.3f -> %.3f, ctc %.2f -> %.2f  (%.1f s)' % (
        first[0], last[0], first[1], last[1], first[2], last[2], time.perf_counter() - t0))
    assert last[0] < 0.5 * first[0] and last[1] < 0.8 * first[1] and last[2] < 0.5 * first[2], 'the losses did not fall'
    print('train sanity ok')


if __name__ == '__main__':
    main()
