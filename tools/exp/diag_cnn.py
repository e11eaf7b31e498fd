"""bf16 conv front end at model level: gradient errors against the oracle with the fast paths and with the general kernels."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import neurst_amd; neurst_amd.want_hw_queues()
import test_gpu_model as T
from neurst_amd import ops
ge = T.ge
for fast in (1, 0):
    ops.set_option('conv_fast', fast)
    args = ge.tiny_args(compute_dtype='bf16', loss_type='ce_ctc', use_cnn_resample=True, cnn={'layers': 2, 'kernel': '3,3', 'stride': '2,2'}, num_filters=32)
    model = T.make_model(args)
    P = T.perturb(model)
    P = {k: (torch.from_numpy(v).bfloat16().float().numpy() if v.ndim > 1 and 'embedding' not in k else v) for k, v in P.items()}
    model.store.load_params(P)
    batch = T.make_batch(B=2, Traw=37, V=11, L=5, U=3, raw=False, dim_input=24, ragged=True, seed=5)
    out4 = model.forward_backward(batch).cpu().numpy()
    cfg = ge.oracle_cfg_from_args(args).zero_dropout()
    ref = T.OracleTPE(cfg, is_train=True, dropout=False).loss_and_grads({k: v.astype(np.float64) for k, v in P.items()}, batch)
    g = model.store.export_grads()
    errs = sorted(((np.abs(g[k] - v).max() / max(np.abs(v).max(), 1e-7), k) for k, v in ref['grads'].items()), reverse=True)
    print('fast', fast, 'loss', out4[0], ref['loss'], [(round(e, 3), k.split('transformer/')[-1]) for e, k in errs[:6]])
    print('   conv:', [(round(e, 3), k.split('transformer/')[-1]) for e, k in errs if 'conv' in k])
