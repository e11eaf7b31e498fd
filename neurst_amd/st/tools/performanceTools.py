"""st/tools/performanceTools.py:10-132 - the dev loop: decode every batch with the inference model, accumulate
CER (token ids) and WER (detokenised words), write the result / reference text files and score BLEU the way
multi-bleu.perl does (st/tools/bleu.py; the reference shells out to perl, :87-100)."""
import logging
import os
import subprocess
from time import time

import numpy as np

from .bleu import bleu_from_files
from .textTools import array2text, batch_cer, batch_wer

_CTC_LOSSES = ('ctc_kd_ce', 'CTC_BERT', 'ce_ctc')


def _np(x):
    return x.numpy() if hasattr(x, 'numpy') else np.asarray(x)


def dev_test(args, step, dataloader, model, sess=None, unit=None, idx2token=None, eos_idx=None, min_idx=0, max_idx=None, out_dir='.'):
    """-> (cer, wer, bleu).  `dataloader` yields the batch dicts of TFReader (mode != 'train': one pass); `model` is the
    inference model (`infer(features, len_feas) -> (preds, preds_am)`); `sess` is kept for the signature only."""
    start_time, batch_time = time(), time()
    processed = 0
    total_cer_dist = total_cer_len = total_wer_dist = total_wer_len = 0
    total = len(dataloader) if hasattr(dataloader, '__len__') else None
    path = lambda name: os.path.join(out_dir, name)
    text = lambda ids, table, max_idx_, delimiter=None: ''.join(
        array2text(ids, args.data.unit, table, min_idx=0, max_idx=max_idx_, args=args, delimiter=delimiter)).replace('<eos>', '') + '\n'
    with open(path('dev_wav.txt'), 'w'), open(path('dev_ast_res.txt'), 'w') as fastres, open(path('dev_ast_ref.txt'), 'w') as fastref, \
            open(path('dev_asr_res.txt'), 'w') as fasrres, open(path('dev_asr_ref.txt'), 'w') as fasrref:
        for batch in dataloader:
            if not batch:
                continue
            decoded, decoded_asr = model.infer(batch['feature'], batch['len_fea'])
            shape_batch = model.last['shape_batch']
            label = _np(batch['label'])
            batch_cer_dist, batch_cer_len = batch_cer(result=decoded, reference=label, eos_idx=eos_idx, min_idx=min_idx, max_idx=max_idx)
            total_cer_dist += batch_cer_dist
            total_cer_len += batch_cer_len
            batch_wer_dist, batch_wer_len = batch_wer(result=decoded, reference=label, idx2token=idx2token, unit=unit, eos_idx=eos_idx,
                                                      min_idx=min_idx, max_idx=max_idx, args=args)
            total_wer_dist += batch_wer_dist
            total_wer_len += batch_wer_len
            used_time = time() - batch_time
            batch_time = time()
            processed += shape_batch[0]
            logging.info('batch cer: {:.3f}\twer: {:.3f} batch: {}\t time:{:.2f}s {}'.format(
                batch_cer_dist / max(batch_cer_len, 1), batch_wer_dist / max(batch_wer_len, 1), shape_batch, used_time,
                '{:.3f}%'.format(processed / total * 100.0) if total else ''))
            for id_line in decoded:
                fastres.write(text(id_line, args.idx2token, args.dim_output - 1))
            for id_line in label:
                fastref.write(text(id_line, args.idx2token, args.dim_output - 1))
            if args.model.loss_type in _CTC_LOSSES:
                src = bool(args.dirs and args.dirs.src_vocab)
                table, top = (args.src_idx2token, args.src_dim_output - 1) if src else (args.idx2token, args.dim_output - 1)
                delim = args.data.src_delimiter if src else None
                for id_line in decoded_asr:
                    fasrres.write(text(id_line, table, top, delim))
                for id_line in _np(batch['aux_label']):
                    fasrref.write(text(id_line, table, top, delim))
    used_time = time() - start_time
    cer = total_cer_dist / total_cer_len
    wer = total_wer_dist / total_wer_len
    if args.bleu_cmd:      # a user-supplied command keeps working as in the reference (:88-95)
        bleu = subprocess.getoutput(args.bleu_cmd.strip().format(ref=path('dev_ast_ref.txt'), output=path('dev_ast_res.txt')))
    else:
        try:
            bleu = bleu_from_files(path('dev_ast_ref.txt'), path('dev_ast_res.txt'))
        except ZeroDivisionError:      # perl dies on an empty output; the reference then reports 0 (:97-100)
            bleu = 0
    try:
        bleu = float(bleu)
    except Exception:
        bleu = 0
    logging.info('=====dev info, total used time {:.2f}h==== \nWER: {:.4f}\ntotal_wer_len: {}\nBLEU: {:.4f}'.format(
        used_time / 3600, wer, total_wer_len, bleu))
    return cer, wer, bleu


def decode_test(step, sample, model, sess=None, unit=None, idx2token=None, eos_idx=None, min_idx=0, max_idx=None, args=None):
    """:110-132 - decode one sample and log result / reference text"""
    feature = np.expand_dims(sample['feature'], axis=0)
    sampled_id, _ = model.infer(feature, np.array([len(sample['feature'])], np.int32))
    res_txt = array2text(sampled_id[0], unit, idx2token, eos_idx, min_idx, max_idx, args=args)
    ref_txt = array2text(sample['label'], unit, idx2token, eos_idx, min_idx, max_idx, args=args)
    logging.info('length: {}, res: \n{}\nref: \n{}'.format(model.last['shape_batch'][1], res_txt, ref_txt))
    return res_txt, ref_txt
