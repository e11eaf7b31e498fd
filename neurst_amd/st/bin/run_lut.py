#!/usr/bin/env python
"""The caller of the hot path: the train / infer loops of st/bin/run_lut.py (:30-455 train, :457-598 infer) and
st/bin/run_costt.py (same loops, multilabel batches) without the TF session plumbing.  One process per GPU; with
WORLD_SIZE > 1 (torchrun) every rank reads its own shard of the batches and gradients are all-reduced over RCCL.

    python -m neurst_amd.st.bin.run_lut -m train -c configs/lut.yaml [--vocab-size 10000]

YAML keys used here, as in the reference: dirs.{train,dev}.tfdata, dir_model / dir_log / dir_checkpoint, num_epochs,
num_steps, dev_step, begin_dev_step, early_stop, metrics (BLEU | WER), data.train.size_dataset, data.unit."""
import argparse
import logging
import os
import sys
from datetime import datetime
from time import time

import numpy as np

from ..layers.arguments import build_args
from ..models.TransformerTPE import GlobalStep
from ..tools.performanceTools import dev_test
from ..tools.summaryTools import Summary
from ..tools.tftools import checkpointTools
from ..tools.tftools.tfRecord import Prefetcher, TFReader


def _rank_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


class _Shard:
    """rank r of N takes batches r, r + N, ... of the common, deterministic batch stream"""

    def __init__(self, it, rank, world):
        self.it, self.rank, self.world = iter(it), rank, world

    def __iter__(self):
        return self

    def __next__(self):
        for i in range(self.world):
            b = next(self.it)
            if i == self.rank:
                keep = b
        return keep


class _Control:
    """Control-plane messages between the ranks (the early-stop decision after a dev pass).  They do not travel over the
    gradient process group: while rank 0 decodes the dev set the other ranks would sit inside an RCCL collective and its
    watchdog would end the job when the pass outlasts the process-group timeout.  A second, gloo group with a day-long
    timeout carries one host integer instead; the waiting ranks hold no device collective meanwhile."""

    def __init__(self, world):
        self.world = world
        self.group = None
        if world > 1:
            import datetime
            import torch.distributed as dist
            self.group = dist.new_group(backend='gloo', timeout=datetime.timedelta(hours=24))

    def broadcast_flag(self, flag):
        """rank 0's decision, on every rank"""
        if self.group is None:
            return bool(flag)
        import torch
        import torch.distributed as dist
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.broadcast(t, src=0, group=self.group)
        return bool(int(t[0]))


def build_models(args):
    gs = GlobalStep(0)
    kw = dict(encoder=args.model.encoder.type, decoder=args.model.decoder.type, args=args)
    model = args.Model(gs, is_train=True, **kw)
    model_infer = args.Model(gs, is_train=False, **kw)        # shares the variables by name (:128-134)
    return model, model_infer


def train(args, max_steps=None):
    """:30-455.  Returns the list of per-step losses (for tests)."""
    rank, world = _rank_world()
    control = _Control(world)
    model, model_infer = build_models(args)
    os.makedirs(args.dir_checkpoint, exist_ok=True)
    summary = Summary(args.dir_log) if rank == 0 else None
    if args.dirs.checkpoint_init:                           # :140-276 restore before training
        checkpointTools.restore(args.dirs.checkpoint_init, model.store)
    reader = TFReader(args.dirs.train.tfdata, args, is_train=True, pin_memory=True, mode='train')
    batches = Prefetcher(_Shard(reader, rank, world), depth=4)
    size_dataset = (args.data.train.size_dataset if args.data.train else None) or float('inf')
    global_bleu, global_wer, bad_count = 0.0, float('inf'), 0

    def evaluate_decode(global_step):
        dev = TFReader(args.dirs.dev.tfdata, args, is_train=False, mode='infer')
        dev_cer, dev_wer, dev_bleu = dev_test(args, global_step, dev, model_infer, None, args.data.unit, args.idx2token, eos_idx=args.eos_idx,
                                              min_idx=0, max_idx=args.dim_output - 1, out_dir=args.dir_log)
        for tag, v in (('dev_cer', dev_cer), ('dev_wer', dev_wer), ('dev_bleu', dev_bleu)):
            summary.summary_scalar(tag, v, global_step)
        return dev_cer, dev_wer, dev_bleu

    def save(global_step):
        # checkpoint_format: 'npz' (default) or 'tf' = TensorBundle files a TensorFlow reader opens (build-side YAML key)
        path = checkpointTools.save_checkpoint('{}/model-{}'.format(args.dir_checkpoint, global_step), model.store, global_step,
                                               fmt=args.checkpoint_format)
        logging.info('Save model in %s.' % path)

    def check_save(dev_bleu, dev_wer, global_step):
        nonlocal global_bleu, global_wer, bad_count
        better = dev_bleu >= global_bleu if args.metrics == 'BLEU' else dev_wer <= global_wer
        if better:
            global_bleu, global_wer = (dev_bleu, global_wer) if args.metrics == 'BLEU' else (global_bleu, dev_wer)
            save(global_step)
        else:
            logging.info('Model not saved, best %s.' % ('bleu is %s' % global_bleu if args.metrics == 'BLEU' else 'wer is %s' % global_wer))
            bad_count += 1

    def dev_and_maybe_stop(global_step):
        """every rank calls this at the same steps: rank 0 scores the dev set, all ranks learn its stop decision"""
        if not (args.dirs.dev and args.dirs.dev.tfdata):
            return False
        stop = False
        if rank == 0:
            _, dev_wer, dev_bleu = evaluate_decode(global_step)
            check_save(dev_bleu, dev_wer, global_step)
            stop = bool(args.early_stop and bad_count > args.early_stop)
        return control.broadcast_flag(stop)

    start_time = datetime.now()
    losses, num_processed, progress = [], 0, 0.0
    dev_and_maybe_stop(0)                                    # :339-341
    while progress < (args.num_epochs or float('inf')):
        batch_time = time()
        global_step = model.global_step.value
        loss, shape_batch, loss_ast, loss_ctc, loss_bert = model.train_step(next(batches))
        lr = model.last['lr']
        used_time = time() - batch_time
        num_processed += shape_batch[0] * world
        progress = num_processed / size_dataset
        losses.append(loss)
        logging.info('loss:{:.3f}\tloss_ast:{:.3f}\tloss_ctc:{:.3f}\tloss_bert:{:.3f}\tbatch:{}\tlr:{:.6f}\ttime:{:.2f}s\tprocess:{:.3f}%\tstep:{}'.format(
            loss, loss_ast, loss_ctc, loss_bert, shape_batch, lr, used_time, progress * 100.0, global_step))
        if summary:
            for tag, v in (('loss_ast', loss_ast), ('loss_ctc', loss_ctc), ('loss_bert', loss_bert), ('loss', loss), ('learning_rate', lr)):
                summary.summary_scalar(tag, v, global_step)
        if args.begin_dev_step and global_step < args.begin_dev_step:          # :416-431
            if global_step % 10000 == 0 and global_step != 0:
                if dev_and_maybe_stop(global_step):
                    logging.info('Model training is early stopped!')
                    break
            elif global_step % 1000 == 0 and global_step != 0 and rank == 0:
                save(global_step)
        elif args.dev_step and args.dev_step > 0 and global_step % args.dev_step == 0 and global_step != 0:
            if dev_and_maybe_stop(global_step):
                logging.info('Model training is early stopped!')
                break
        elif not args.dev_step and global_step % 1000 == 0 and global_step != 0 and rank == 0:
            save(global_step)
        if (args.num_steps and global_step > args.num_steps) or (max_steps and len(losses) >= max_steps):
            break
    batches.close()
    logging.info('training duration: {:.2f}h'.format((datetime.now() - start_time).total_seconds() / 3600))
    return losses


def infer(args):
    """:457-598 - decode the test (or dev) records with the restored model and score them"""
    _, model_infer = build_models(args)
    ckpt = args.dirs.checkpoint or args.dir_checkpoint
    step = checkpointTools.restore(ckpt, model_infer.store, slots=False)
    data = (args.dirs.test and args.dirs.test.tfdata) or args.dirs.dev.tfdata
    reader = TFReader(data, args, is_train=False, mode='infer')
    cer, wer, bleu = dev_test(args, step, reader, model_infer, None, args.data.unit, args.idx2token, eos_idx=args.eos_idx, min_idx=0,
                              max_idx=args.dim_output - 1, out_dir=args.dir_log or '.')
    logging.info('infer CER: {:.4f} WER: {:.4f} BLEU: {:.2f}'.format(cer, wer, bleu))
    return cer, wer, bleu


def load_vocab(path):
    """st/tools/vocab.py / arguments.py:92 - one token per line (first column), index = line number"""
    token2idx = {}
    with open(path, encoding='utf-8') as f:
        for line in f:
            if line.strip():
                token2idx.setdefault(line.split()[0], len(token2idx))
    return token2idx, {i: t for t, i in token2idx.items()}


def main(argv=None):
    from ... import want_hw_queues
    want_hw_queues()      # before the HIP runtime starts: the step keeps 3-4 streams busy (neurst_amd/__init__.py)
    logging.basicConfig(level=logging.INFO, format='%(levelname)s(%(filename)s:%(lineno)d): %(message)s')
    parser = argparse.ArgumentParser()
    parser.add_argument('-m', type=str, dest='mode', default='train')
    parser.add_argument('--name', type=str, dest='name', default=None)
    parser.add_argument('--gpu', type=str, dest='gpu', default='0')
    parser.add_argument('-c', type=str, dest='config', nargs='+', help='Configuration files.')
    parser.add_argument('--vocab-size', type=int, default=None, help='used when dirs.vocab is absent (synthetic ids)')
    param = parser.parse_args(argv)
    from ..layers.arguments import load_config
    cfg = load_config(param.config)
    vocab = (cfg.get('dirs') or {}).get('vocab')
    token2idx, idx2token = load_vocab(vocab) if vocab else (None, None)
    args = build_args(cfg, dim_output=len(token2idx) if token2idx else param.vocab_size, token2idx=token2idx)
    args.idx2token = idx2token or {i: str(i) for i in range(args.dim_output)}
    args.mode = param.mode
    if int(os.environ.get('WORLD_SIZE', 1)) > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
        dist.init_process_group('nccl')
    if param.mode == 'train':
        train(args)
    elif param.mode == 'infer':
        infer(args)
    else:
        raise NotImplementedError('mode {}: only train and infer are on the path'.format(param.mode))


if __name__ == '__main__':
    main(sys.argv[1:])
