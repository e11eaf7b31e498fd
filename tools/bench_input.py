#!/usr/bin/env python
"""Input pipeline measurement: synthetic `*.recode` files of BASELINE config 2's shape (80-dim raw
frames, ~1000-1500 per utterance, 40 labels, 40 x 768 BERT features) are written once, then
  (1) the native scanner's CRC-32C pass,                                   GB/s
  (2) TFReader on one host thread (views + padding copies into pinned memory), raw frames/s against what one GPU consumes,
  (3) with a GPU: LUT train steps fed by TFReader + Prefetcher against the same steps on one resident batch.
Usage: python tools/bench_input.py [--utts 640] [--steps 20] [--dir /tmp/neurst_records]"""
import argparse
import json
import os
import shutil
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neurst_amd.st.layers.arguments import build_args  # noqa: E402
from neurst_amd.st.tools.tftools import tfRecord as R  # noqa: E402


def samples(n, seed=0):
    rng = np.random.default_rng(seed)
    for i in range(n):
        T = int(rng.integers(1000, 1501))
        lab = rng.integers(4, 9998, size=40).astype(np.int32)
        lab[-1] = 3
        yield {'id': np.array([i], np.int32), 'feature': rng.standard_normal((T, 80), dtype=np.float32), 'label': lab, 'aux_label': lab,
               'bert_feature': rng.standard_normal((40, 768), dtype=np.float32)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--utts', type=int, default=640)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--dir', default='/tmp/neurst_records')
    a = ap.parse_args()
    shutil.rmtree(a.dir, ignore_errors=True)
    t0 = time.perf_counter()
    R.save2tfrecord_ctcbert(samples(a.utts), a.dir, size_file=200000)
    files = R.fentch_filelist(a.dir)
    nbytes = sum(os.path.getsize(f) for f in files)
    out = {'files': len(files), 'MB': round(nbytes / 1e6, 1), 'write_s': round(time.perf_counter() - t0, 2)}

    for f in files:                       # page cache warm
        R.RecordFile(f, verify_crc=False)
    t0 = time.perf_counter()
    for f in files:
        R.RecordFile(f, verify_crc=True)
    out['crc_scan_GB_per_s'] = round(nbytes / (time.perf_counter() - t0) / 1e9, 2)

    args = build_args(os.path.join(ROOT, 'configs', 'lut.yaml'), dim_output=10000, compute_dtype='bf16')
    args.mode, args.num_batch_tokens, args.bucket_boundaries, args.list_bucket_boundaries = 'train', 16000, '400,501', None
    args.data.max_train_len = 1500
    for pin in (False, True):
        if pin and not torch.cuda.is_available():
            continue
        it = iter(R.TFReader(a.dir, args, is_train=True, pin_memory=pin, verify_crc=False))
        next(it)
        frames, t0 = 0, time.perf_counter()
        for _ in range(20):
            frames += int(next(it)['len_fea'].sum())
        out['reader_raw_frames_per_s' + ('_pinned' if pin else '')] = round(frames / (time.perf_counter() - t0))

    if torch.cuda.is_available():
        from neurst_amd.st.models.TransformerTPE import GlobalStep
        model = args.Model(GlobalStep(1000), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
        pre = R.Prefetcher(R.TFReader(a.dir, args, is_train=True, pin_memory=True, verify_crc=True), depth=4)
        first = next(pre)
        resident = model._to_device(first)
        for _ in range(3):
            model.train_step_async(resident)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            model.train_step_async(resident)
        torch.cuda.synchronize()
        dt_res = (time.perf_counter() - t0) / a.steps
        enc = int((-(-first['len_fea'] // 3)).sum())
        out['resident_batch'] = {'B': int(first['feature'].shape[0]), 'ms_per_step': round(dt_res * 1e3, 2), 'enc_frames_per_s': round(enc / dt_res)}
        for _ in range(3):
            model.train_step_async(next(pre))
        torch.cuda.synchronize()
        enc, t0 = 0, time.perf_counter()
        for _ in range(a.steps):
            b = next(pre)
            enc += int((-(-b['len_fea'] // 3)).sum())
            vals = model.train_step_async(b)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.isfinite(vals).all()
        out['from_records'] = {'ms_per_step': round(dt / a.steps * 1e3, 2), 'enc_frames_per_s': round(enc / dt), 'raw_frames_per_s': round(3 * enc / dt)}
        pre.close()
    print(json.dumps(out), flush=True)
    shutil.rmtree(a.dir, ignore_errors=True)


if __name__ == '__main__':
    main()
