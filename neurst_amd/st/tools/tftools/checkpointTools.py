"""Checkpoints of the variable store under the reference's TF variable names and layouts - SURVEY 8(f) rank 1.

Mirrors the reading interface of st/tools/tftools/checkpointTools.py:44-97 (`load_checkpoint(...).get_tensor`,
`load_variable`, `list_variables`), what `tf.train.Saver` keeps for this model (`st/bin/run_lut.py:140-276`: variables,
Adam slots `<var>/Adam`, `<var>/Adam_1`, `beta1_power`, `beta2_power`, `global_step`) and st/tools/avg_checkpoints.py:46-117.

Two containers, told apart by the files present: TensorFlow's own TensorBundle (`<prefix>.index` + `<prefix>.data-*`,
tensorBundle.py - what a reference-trained model is saved as) and ONE numpy `.npz` (plain arrays, read with
allow_pickle=False; the default for writing).  Variable names and array layouts are the TF ones in both (`[out, in]` for
`dense`, `[in, out]` for `tf.layers.dense` kernels).  A directory stands for its newest checkpoint as named by its
`checkpoint` file - TF's text format (`model_checkpoint_path: "model-1000"`) or one file name per line."""
import os

import numpy as np

from .tensorBundle import BundleReader, write_bundle

ADAM_M, ADAM_V = '/Adam', '/Adam_1'


class CheckpointReader:
    """Subset of pywrap_tensorflow.CheckpointReader used by the reference (get_tensor, get_variable_to_shape_map)."""

    def __init__(self, path):
        self.path = resolve(path)
        if self.path.endswith('.npz'):
            self._z, self._b = np.load(self.path, allow_pickle=False), None
        else:
            self._z, self._b = None, BundleReader(self.path)

    def get_tensor(self, name):
        name = name[:-2] if name.endswith(':0') else name
        return np.array(self._z[name]) if self._b is None else self._b.get_tensor(name)

    def has_tensor(self, name):
        return name in self._z.files if self._b is None else self._b.has_tensor(name)

    def get_variable_to_shape_map(self):
        return {k: list(self._z[k].shape) for k in self._z.files} if self._b is None else self._b.get_variable_to_shape_map()


def resolve(ckpt_dir_or_file):
    """-> `<...>.npz`, or the prefix of a TensorBundle.  A directory means its newest checkpoint
    (tf.train.latest_checkpoint)."""
    p = ckpt_dir_or_file
    if os.path.isdir(p):
        index = os.path.join(p, 'checkpoint')
        if not os.path.exists(index):
            raise ValueError("Couldn't find 'checkpoint' file or checkpoints in given directory %s" % p)
        lines = [l.strip() for l in open(index) if l.strip()]
        newest = [l.split(':', 1)[1].strip().strip('"') for l in lines if l.startswith('model_checkpoint_path:')]
        p = os.path.join(p, newest[-1] if newest else lines[-1])
    if p.endswith('.npz') or os.path.exists(p + '.index'):
        return p
    return p + '.npz'


def _note_in_state_file(full):
    """appends the new checkpoint to `<dir>/checkpoint` in the format that file already has (TF's for bundles)"""
    d, name = os.path.dirname(os.path.abspath(full)), os.path.basename(full)
    state = os.path.join(d, 'checkpoint')
    if full.endswith('.npz'):
        with open(state, 'a') as f:
            f.write(name + '\n')
        return
    older = []
    if os.path.exists(state):
        older = [l.split(':', 1)[1].strip().strip('"') for l in open(state) if l.startswith('all_model_checkpoint_paths:')]
    with open(state, 'w') as f:
        f.write('model_checkpoint_path: "%s"\n' % name)
        for n in [o for o in older if o != name] + [name]:
            f.write('all_model_checkpoint_paths: "%s"\n' % n)


def _write(path, tensors, fmt):
    if fmt not in (None, 'npz', 'tf'):
        raise ValueError("checkpoint format must be 'npz' or 'tf'")
    if fmt == 'tf' or (fmt is None and not path.endswith('.npz') and os.path.exists(path + '.index')):
        full = path[:-4] if path.endswith('.npz') else path
        write_bundle(full, tensors)
    else:
        full = path if path.endswith('.npz') else path + '.npz'
        os.makedirs(os.path.dirname(os.path.abspath(full)), exist_ok=True)
        np.savez(full, **tensors)
    return full


def load_checkpoint(ckpt_dir_or_file):
    return CheckpointReader(ckpt_dir_or_file)


def load_variable(ckpt_dir_or_file, name):
    return load_checkpoint(ckpt_dir_or_file).get_tensor(name)


def list_variables(ckpt_dir_or_file):
    m = load_checkpoint(ckpt_dir_or_file).get_variable_to_shape_map()
    return [(k, m[k]) for k in sorted(m)]


def save_checkpoint(path, store, global_step, beta1=0.9, beta2=0.98, slots=True, fmt=None):
    """Writes variables, Adam slots, beta powers and global_step as `<path>.npz` (default) or, with fmt='tf', as the
    TensorBundle `<path>.index` / `<path>.data-00000-of-00001`, and notes it in `<dir>/checkpoint`."""
    out = store.export()
    if slots:
        for k, v in store.export(store.m).items():
            out[k + ADAM_M] = v
        for k, v in store.export(store.v).items():
            out[k + ADAM_V] = v
        out['beta1_power'] = np.float32(beta1 ** max(store.adam_t, 0) * beta1)   # TF keeps beta^(t+1)
        out['beta2_power'] = np.float32(beta2 ** max(store.adam_t, 0) * beta2)
        out['adam_t'] = np.int64(store.adam_t)
    out['global_step'] = np.int64(int(global_step))
    full = _write(path, out, fmt)
    _note_in_state_file(full)
    return full


def restore(ckpt_dir_or_file, store, assignment_map=None, slots=True, strict=True):
    """Loads variables (and Adam slots) into the store; returns the checkpoint's global_step.
    assignment_map {'checkpoint_scope/': 'scope/'} renames prefixes as `init_from_checkpoint` does (:100-160)."""
    r = load_checkpoint(ckpt_dir_or_file)

    def ckpt_name(var):
        for src, dst in (assignment_map or {}).items():
            dst = '' if dst == '/' else dst
            src = '' if src == '/' else src
            if var.startswith(dst):
                return src + var[len(dst):]
        return var

    params, m, v, missing = {}, {}, {}, []
    for name, p in store.vars.items():
        cn = ckpt_name(name)
        if not r.has_tensor(cn):
            missing.append(name)
            continue
        arr = r.get_tensor(cn)
        if tuple(arr.shape) != p.shape:
            raise ValueError('Shape of variable %s (%s) does not match the checkpoint tensor %s (%s)' % (name, p.shape, cn, arr.shape))
        params[name] = arr
        if slots and r.has_tensor(cn + ADAM_M) and r.has_tensor(cn + ADAM_V):
            m[name], v[name] = r.get_tensor(cn + ADAM_M), r.get_tensor(cn + ADAM_V)
    if missing and strict:
        raise ValueError('variables missing from %s: %s' % (r.path, ', '.join(missing[:5]) + (' ...' if len(missing) > 5 else '')))
    store.load_params(params)
    if slots and m:
        store.load_slots(m, v)
        if r.has_tensor('adam_t'):
            store.adam_t = int(r.get_tensor('adam_t'))
        elif r.has_tensor('beta1_power') or r.has_tensor('beta2_power'):
            store.adam_t = _adam_t_from_powers(r)
    return int(r.get_tensor('global_step')) if r.has_tensor('global_step') else 0


def _adam_t_from_powers(r):
    """Adam's step count of a TensorFlow-written checkpoint, which stores beta1^(t+1) and beta2^(t+1) as fp32 scalars instead
    (lstmModel.py:526-530: beta1 = 0.9, beta2 = 0.98).  0.9^t underflows to 0 in fp32 near t = 1000 and loses digits long
    before; 0.98^t lasts to about t = 4000.  Take whichever power is still well inside the normal range, beta2's first; past
    both, the bias corrections are 1 to fp32 precision and the checkpoint's global_step (one Adam step per train step) is
    the count."""
    tiny = 1e-30
    for name, beta in (('beta2_power', 0.98), ('beta1_power', 0.9)):
        if r.has_tensor(name):
            x = float(r.get_tensor(name))
            if tiny < x < 1.0:
                return max(int(round(np.log(x) / np.log(beta))) - 1, 0)
            if x >= 1.0:
                return 0
    return int(r.get_tensor('global_step')) if r.has_tensor('global_step') else 0


def average_checkpoints(checkpoints, output_path, fmt=None):
    """st/tools/avg_checkpoints.py:75-117: element-wise mean of every variable except global_step (Adam slots included,
    as the reference's list_variables loop does); the result carries global_step 0."""
    checkpoints = [c for c in checkpoints if os.path.exists(resolve(c)) or os.path.exists(resolve(c) + '.index')]
    if not checkpoints:
        raise ValueError('None of the provided checkpoints exist.')
    names = [n for n, _ in list_variables(checkpoints[0]) if not n.startswith('global_step') and n != 'adam_t']
    acc = {}
    for c in checkpoints:
        r = load_checkpoint(c)
        for n in names:
            t = r.get_tensor(n).astype(np.float64)
            acc[n] = t if n not in acc else acc[n] + t
    out = {n: (a / len(checkpoints)).astype(np.float32) for n, a in acc.items()}
    out['global_step'] = np.int64(0)
    return _write(output_path, out, fmt)
