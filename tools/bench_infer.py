#!/usr/bin/env python
"""Inference at BASELINE config 2's width (LUT: d=768, 4+4+4 blocks, V=10000) on one MI355X: greedy decoder_with_caching
and beam_decode_rerank (transformer_decoder.py:52-182, :248-431) on B utterances of Traw raw frames, random weights
(no hypothesis reaches <eos>, so every run decodes exactly --max-len steps).  Prints the time of the whole infer() call
(encoder + decoding + read-back) and of one decoding step.
Usage: python tools/bench_infer.py [--beam 1 4] [--batch 32] [--traw 1500] [--max-len 100] [--length-penalty 0.0]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neurst_amd.st.layers.arguments import build_args  # noqa: E402
from neurst_amd.st.models.TransformerTPE import GlobalStep  # noqa: E402
from neurst_amd.synthetic import make_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--beam', type=int, nargs='+', default=[1, 4])
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--traw', type=int, default=1500)
    ap.add_argument('--max-len', type=int, default=100)
    ap.add_argument('--length-penalty', type=float, default=0.0)
    ap.add_argument('--config', default='lut.yaml')
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    host = make_batch(B=a.batch, Traw=a.traw, V=10000, L=40, U=40, Lb=40, bert_dim=None, seed=6, ragged=True)
    for beam in a.beam:
        args = build_args(os.path.join(ROOT, 'configs', a.config), dim_output=10000, compute_dtype='bf16')
        args.beam_size, args.length_penalty_weight = beam, a.length_penalty
        times = {}
        for max_len in (a.max_len // 2, a.max_len):       # two lengths: the difference is pure decoding
            args.max_len = max_len
            model = args.Model(GlobalStep(0), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=False, args=args)
            model.infer(host['feature'], host['len_fea'])
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(a.reps):
                t0 = time.perf_counter()
                preds, _ = model.infer(host['feature'], host['len_fea'])
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            assert preds.shape == (a.batch, max_len), preds.shape
            times[max_len] = best
            del model
        per_step = (times[a.max_len] - times[a.max_len // 2]) / (a.max_len - a.max_len // 2)
        print(json.dumps({'workload': '%s infer B=%d Traw=%d beam=%d max_len=%d' % (a.config, a.batch, a.traw, beam, a.max_len),
                          'infer_ms': round(times[a.max_len] * 1e3, 2), 'ms_per_decoding_step': round(per_step * 1e3, 3),
                          'hypothesis_tokens_per_s': round(a.batch * beam / per_step), 'utterance_tokens_per_s': round(a.batch / per_step),
                          'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2**30, 2)}), flush=True)


if __name__ == '__main__':
    main()
