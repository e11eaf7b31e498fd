"""LSTM_Model base of st/models/lstmModel.py: the parts every model class inherits on the hot path -
constructor bookkeeping (:17-60) and build_optimizer (:487-538: warmup_exponential_decay LR and
tf.train.AdamOptimizer with beta1=0.9, beta2=0.98, epsilon=1e-9 hard-coded, YAML `opti` ignored)."""
import math

from .model_tools import warmup_exponential_decay
from ... import ops as _ops


class LSTM_Model(object):
    num_Instances = 0
    num_Model = 0

    ADAM_BETA1, ADAM_BETA2, ADAM_EPS = 0.9, 0.98, 1e-9

    def __init__(self, tensor_global_step, is_train, args, batch=None, name='model'):
        self.is_train = is_train
        self.num_gpus = args.num_gpus if is_train else 1
        self.list_gpu_devices = args.list_gpus
        self.center_device = '/cpu:0'
        self.args = args
        self.batch = batch
        self.name = name
        self.list_pl = None
        self.global_step = tensor_global_step
        self.list_run = list(self.build_graph() if is_train else self.build_infer_graph())

    def lr_at(self, global_step):
        a = self.args
        if a.lr_type and a.lr_type != 'warmup_exponential_decay':
            raise NotImplementedError('lr_type {} is outside the hot path (both templates use the default)'.format(a.lr_type))
        return warmup_exponential_decay(global_step, warmup_steps=a.warmup_steps, peak=a.peak, decay_rate=0.5, decay_steps=a.decay_steps)

    @property
    def learning_rate(self):
        """LR of the NEXT step = f(global_step before increment), as `sess.run([global_step, lr])` reads it (run_lut.py:346)."""
        return self.lr_at(self.global_step.value)

    def build_optimizer(self):
        if self.args.optimizer != 'adam':
            raise NotImplementedError('only optimizer: adam is on the hot path')
        return self

    def optimizer_update_fn(self, store, grad_scale, skip_flag=None):
        """-> (lr, fn): fn(start, end) enqueues this step's Adam update of arena[start:end] on the current stream
        (used to update gradient buckets while backward is still running); finish_update() closes the step."""
        lr = self.lr_at(self.global_step.value)
        t = store.adam_t + 1
        b1, b2 = self.ADAM_BETA1, self.ADAM_BETA2
        lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)

        def fn(s, e):
            _ops.adam_step(store.params[s:e], store.grads[s:e], store.m[s:e], store.v[s:e],
                           None if store.shadow is None else store.shadow[s:e], lr_t, b1, b2, self.ADAM_EPS, grad_scale, skip_flag=skip_flag)
        return lr, fn

    def finish_update(self, store):
        store.adam_t += 1
        self.global_step.value += 1

    def apply_gradients(self, store, grad_scale, skip=(), skip_flag=None):
        """TF Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v update; theta -= lr_t*m/(sqrt(v)+eps); global_step += 1.
        `skip`: arena ranges (start, end) of variables that are not handed to the optimizer (frozen_scope,
        TransformerTPE.py:171-183): their values and Adam slots stay bit-identical."""
        lr, fn = self.optimizer_update_fn(store, grad_scale, skip_flag=skip_flag)
        start = 0
        for s, e in sorted(skip):
            if s > start:
                fn(start, s)
            start = max(start, e)
        if start < store.total:
            fn(start, store.total)
        self.finish_update(store)
        return lr
