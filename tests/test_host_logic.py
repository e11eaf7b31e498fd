"""CPU tests of the host side: C ABI exports, config loader, LR schedule, variable naming,
gradient bucketing with a 2-rank gloo group."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from neurst_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        ge.build()
    protos = lib.parse_header()
    text = open(lib.HEADER).read()
    declared = set(re.findall(r'\b(neurst_\w+)\s*\(', re.sub(r'/\*.*?\*/', '', text, flags=re.S)))
    assert declared == set(protos) and len(protos) >= 40
    L = lib.load()
    for name in protos:
        assert hasattr(L, name), name
    assert L.neurst_abi_version() == 3      # == NEURST_ABI_VERSION of include/neurst_hip.h (lib.load() checks the pair)
    # argument validation happens on the host before any launch: callable without a GPU
    assert L.neurst_gemm_bias_act(0, 0, 0, 0, 0, 4, 4, 4, 4, 4, 4, 0, 1, 0, 0, 0, 0, 1, 0) == -1
    assert b'null' in L.neurst_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    from neurst_amd import lib
    monkeypatch.setattr(lib, '_lib', None)
    monkeypatch.setattr(lib, 'LIB_PATH', '/nonexistent/libneurst_hip.so')
    with pytest.raises(lib.NeurstLibError):
        lib.load()


def test_attrdict_and_config_semantics():
    from neurst_amd.st.layers.arguments import AttrDict, build_args
    a = AttrDict({'x': {'y': 1}})
    assert a.x.y == 1 and a.nope is None and a.x.nope is None
    args = build_args(os.path.join(ROOT, 'configs', 'lut.yaml'), dim_output=10000)
    assert args.data.dim_input == 480 and args.num_gpus == 1 and args.model.kd.num_heads == 4
    assert args.list_batch_size[0] == int(10000 / 102) and args.list_batch_size[-1] == 1
    assert args.model.encoder.type.__name__ == 'Transformer_Encoder' and args.Model.__name__ == 'TransformerTPE'
    assert args.model.use_cnn_resample is None and args.model.shrink_layer is None
    c = build_args(os.path.join(ROOT, 'configs', 'costt.yaml'), dim_output=10000)
    assert c.model.shrink_layer is True and c.model.encoder.num_cell_units == 512


@pytest.mark.skipif(not os.path.isdir('/root/reference/configs_template'), reason='reference templates only exist in the build container')
@pytest.mark.parametrize('name', ['lut', 'costt'])
def test_reference_templates_parse_unchanged_and_agree_on_hot_path_keys(name):
    from neurst_amd.st.layers.arguments import build_args
    ref = build_args('/root/reference/configs_template/%s.yaml' % name, dim_output=10000, overrides={'beam_size': 1})
    own = build_args(os.path.join(ROOT, 'configs', '%s.yaml' % name), dim_output=10000)
    for part in ('encoder', 'encoder2', 'decoder'):
        for k in ('num_blocks', 'num_heads', 'num_cell_units', 'attention_dropout_rate', 'residual_dropout_rate'):
            assert getattr(getattr(ref.model, part), k) == getattr(getattr(own.model, part), k)
    for k in ('loss_type', 'shrink_layer', 'use_specAug', 'use_semantic_encoder', 'structure'):
        assert getattr(ref.model, k) == getattr(own.model, k)
    for k in ('warmup_steps', 'peak', 'decay_steps', 'max_len', 'num_batch_tokens'):
        assert getattr(ref, k) == getattr(own, k)
    assert ref.data.dim_input == own.data.dim_input == 480
    from neurst_amd.st.models.TransformerTPE import variable_specs
    assert variable_specs(ref) == variable_specs(own)


def test_variable_names_and_sizes_match_survey_inventory():
    from neurst_amd.st.layers.arguments import build_args
    from neurst_amd.st.models.TransformerTPE import variable_specs
    from oracle.model import TPEConfig, param_specs
    import yaml
    for name, total in (('lut', 120281088), ('costt', 78658560)):
        path = os.path.join(ROOT, 'configs', name + '.yaml')
        specs = variable_specs(build_args(path, dim_output=10000))
        assert sum(int(np.prod(s[1])) for s in specs) == total
        cfg = TPEConfig.from_yaml_dict(yaml.safe_load(open(path)), 10000)
        assert [(a, tuple(b)) for a, b, _ in specs] == [(a, tuple(b)) for a, b, _ in param_specs(cfg)]
        names = [s[0] for s in specs]
        assert 'transformer/decoder/decoder/fully_connected/kernel' in names and 'transformer/embedding' in names
        assert 'transformer/TPEncoder/block_0/ff_hidden/dense_1/kernel' in names


def test_variable_store_is_released_with_its_configuration():
    """The store registry must not keep arenas alive after the configuration they belong to is gone
    (a process that builds several models in turn)."""
    import gc
    from neurst_amd.st.layers.arguments import build_args
    from neurst_amd.st.models import TransformerTPE as M
    args = build_args(os.path.join(ROOT, 'configs', 'lut.yaml'), dim_output=50, compute_dtype='fp32')
    for blk in (args.model.encoder, args.model.encoder2, args.model.decoder):
        blk.num_blocks = 1
    store = M.get_variable_store(args, 'transformer', torch.device('cpu'))
    assert M.get_variable_store(args, 'transformer', torch.device('cpu')) is store      # shared by name (tf.AUTO_REUSE)
    key = (id(args), 'transformer')
    assert key in M._STORES
    del args
    gc.collect()
    assert key not in M._STORES


def test_lr_schedule_matches_oracle():
    from neurst_amd.st.models.model_tools import warmup_exponential_decay as w
    from oracle.ops import warmup_exponential_decay as wo
    for s in (0, 1, 999, 1000, 1001, 50000, 123456):
        assert abs(w(s, 1000, 5e-5, 0.5, 50000) - float(wo(s, 1000, 5e-5, 0.5, 50000))) < 1e-12
    assert w(0, 50000, 4e-4, 0.5, 50000) == 0.0


def test_site_key_matches_oracle():
    from neurst_amd.ops import site_key
    from oracle.ops import site_key as so
    for a in ((0, 0, 0), (17, 523, 99), (2 ** 31, 9000, 10 ** 6)):
        assert site_key(*a) == so(*a)


WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
rank = int(sys.argv[1])
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%%s' %% sys.argv[2], rank=rank, world_size=2)
from neurst_amd.st.tools.tftools.gradientTools import GradBucketer, average_gradients

class P:
    def __init__(s, name, off, size): s.name, s.offset, s.size = name, off, size
class Store:
    pass
st = Store()
sizes = [300, 50, 700, 128, 64, 1000]
offs = np.cumsum([0] + [-(-s // 64) * 64 for s in sizes])
st.vars = {('p%%d' %% i): P('p%%d' %% i, int(offs[i]), sizes[i]) for i in range(len(sizes))}
st.total = int(offs[-1])
st.grads = torch.arange(st.total, dtype=torch.float32) * (rank + 1)
b = GradBucketer(st, bucket_bytes=4 * 512)
assert b.world == 2 and len(b.bounds) > 2
order = ['p5', 'p4', 'p3', 'p1', 'p2']            # backward order; p0 never reports (no gradient this step)
for n in order:
    b.param_done(st.vars[n])
launched_early = sum(b.launched)
scale = average_gradients(b)
want = torch.arange(st.total, dtype=torch.float32) * 3
assert scale == 0.5 and torch.equal(st.grads, want), (scale, (st.grads - want).abs().max())
assert launched_early >= 1 and launched_early < len(b.bounds)
# second step reuses the bucketer
st.grads = torch.ones(st.total) * (rank + 1)
for n in ['p0'] + order:
    b.param_done(st.vars[n])
assert average_gradients(b) == 0.5 and torch.equal(st.grads, torch.ones(st.total) * 3)
# third step: buckets are also UPDATED while "backward" runs - each exactly once, only after its all-reduce, and only
# after the node that completed them has ended (that layer still reads its weights right after reporting)
st.grads = torch.ones(st.total) * (rank + 1)
seen = []
def apply_fn(s, e):
    assert torch.equal(st.grads[s:e], torch.ones(e - s) * 3), 'update before the all-reduce'
    seen.append((s, e))
b.apply_fn = apply_fn
during = []
for n in order:                                   # one "backward node" per variable: param_done, (its dgrad), node_done
    before = len(seen)
    b.param_done(st.vars[n])
    assert len(seen) == before, 'a bucket was updated inside the node that completed it'
    b.node_done()
    during.append(len(seen))
completed = sum(1 for x in b.pending if x == 0)
assert during[-1] == completed and during[-1] >= 1, (during, completed)
assert average_gradients(b) == 0.5
assert sorted(seen) == list(b.bounds), seen                                     # every bucket once, p0's included
b.apply_fn = None
# a variable reported twice is fine while its bucket is untouched, an error afterwards
b.apply_fn = lambda s, e: None
b.param_done(st.vars['p5']); b.param_done(st.vars['p4']); b.param_done(st.vars['p4'])
for n in ['p3', 'p1', 'p2']:
    b.param_done(st.vars[n])
b.node_done()
try:
    b.param_done(st.vars['p4'])
    raise SystemExit('late gradient of a shared variable was accepted')
except RuntimeError:
    pass
average_gradients(b)
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


WORKER2 = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
rank = int(sys.argv[1])
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%%s' %% sys.argv[2], rank=rank, world_size=2)
from neurst_amd.st.tools.tftools.gradientTools import GradBucketer, average_gradients

class P:
    def __init__(s, name, off, size): s.name, s.offset, s.size = name, off, size
class Store:
    pass
# arena in forward order: an "embedding" (complete mid-backward), the acoustic encoder's head (complete LAST), then the rest
names = ['embedding', 'encoder_fc', 'enc0_qkv', 'enc0_ffn', 'enc1', 'ctc_head', 'enc2', 'dec', 'dec_fc']
sizes = [4096, 256, 1024, 3072, 4096, 2048, 4096, 6144, 2048]
offs = np.cumsum([0] + [-(-s // 64) * 64 for s in sizes])
st = Store()
st.vars = {n: P(n, int(offs[i]), sizes[i]) for i, n in enumerate(names)}
st.total = int(offs[-1])
emb_end, tail_end = int(offs[1]), int(offs[3])          # cuts: behind the embedding; behind the first 1280 "last" elements
st.grads = torch.zeros(st.total)
b = GradBucketer(st, bucket_bytes=4 * 4096, cuts=[emb_end, tail_end])
assert (0, emb_end) in b.bounds and (emb_end, tail_end) in b.bounds, b.bounds          # both are buckets of their own
assert all(e - s <= 4096 for s, e in b.bounds) and b.bounds[-1][1] == st.total and all(b.bounds[i][1] == b.bounds[i + 1][0] for i in range(len(b.bounds) - 1))
tail = b.bounds.index((emb_end, tail_end))
backward = ['dec_fc', 'dec', 'embedding', 'enc2', 'ctc_head', 'enc1', 'enc0_ffn', 'enc0_qkv', 'encoder_fc']
base = torch.arange(st.total, dtype=torch.float32) / 64.0
st.grads = base * (rank + 1)
for n in backward:
    b.param_done(st.vars[n])
assert all(b.launched), b.launched                      # every bucket went out during "backward"
assert b.order[-1] == tail, (b.order, tail)             # ... the small head bucket LAST: it is all that is exposed behind backward
assert b.order.index(0) < b.order.index(tail) - 2       # the embedding's bucket left as soon as the decoder was done
assert (b.bounds[tail][1] - b.bounds[tail][0]) * 4 <= 8 * 1024
assert average_gradients(b) == 0.5 and torch.equal(st.grads, base * 3)
# bf16 buckets: same call order; every addend is rounded to bf16 (2^-9 relative) and so is the collective's sum: stated
# tolerance 2^-7 of the sum (three roundings of at most 2^-9 each) - against a 2^-24 fp32 exchange
b16 = GradBucketer(st, bucket_bytes=4 * 4096, cuts=[emb_end, tail_end], bf16=True)
st.grads = base * (rank + 1)
for n in backward:
    b16.param_done(st.vars[n])
assert b16.order == b.order or sorted(b16.order) == sorted(range(len(b16.bounds)))
average_gradients(b16)
err = (st.grads - base * 3).abs()
assert float(err.max()) > 0 and bool((err <= (base * 3).abs() * 2.0 ** -7 + 1e-30).all()), float(err.max())
assert st.grads.dtype == torch.float32
# comm off (bench.py --no-comm): the same bookkeeping and launch order, no collective - gradients stay the local sums
bn = GradBucketer(st, bucket_bytes=4 * 4096, cuts=[emb_end, tail_end])
bn.skip_comm = True
st.grads = base * (rank + 1)
for n in backward:
    bn.param_done(st.vars[n])
assert bn.order[-1] == tail and all(bn.launched)
assert average_gradients(bn) == 0.5 and torch.equal(st.grads, base * (rank + 1))
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


def test_grad_bucketer_cuts_bf16_buckets_and_comm_off_two_rank_gloo(tmp_path):
    """Multi-GPU readiness of a19 (gradientTools.py:7-37) on two gloo ranks: the arena cut so that the gradients that
    complete last form a small bucket that is launched last, the embedding's bucket leaves early; bf16 buckets with
    their stated tolerance; the comm-off leg of the overlap report."""
    script = tmp_path / 'worker2.py'
    script.write_text(WORKER2 % ROOT)
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_grad_bucketer_two_rank_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % ROOT)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


@pytest.mark.parametrize('fmt', [None, 'tf'])
def test_checkpoint_roundtrip_rename_and_average(tmp_path, fmt):
    """checkpoints (.npz, or fmt='tf': TensorBundle .index/.data) under the TF variable names - save / list / load_variable / restore with a
    scope rename / average (st/tools/tftools/checkpointTools.py:44-97, st/tools/avg_checkpoints.py:75-117)."""
    import torch
    from neurst_amd import ops
    from neurst_amd.st.models.graph import VariableStore
    from neurst_amd.st.tools.tftools import checkpointTools as ck
    specs = [('transformer/encoder/encoder_fc/kernel', (6, 4), 'glorot'), ('transformer/encoder/block_0/layer_norm/layer_norm_scale', (4,), 'ones'),
             ('transformer/decoder/decoder_fc/kernel', (4, 7), 'glorot')]
    a = VariableStore(specs, torch.device('cpu'), ops.F32, seed=3)
    a.m += 0.25
    a.v += 0.5
    a.adam_t = 7
    p1 = ck.save_checkpoint(str(tmp_path / 'model-100'), a, 100, fmt=fmt)
    assert os.path.exists(p1 + '.index') == (fmt == 'tf')
    assert [n for n, _ in ck.list_variables(p1) if n.endswith('kernel')] == ['transformer/decoder/decoder_fc/kernel', 'transformer/encoder/encoder_fc/kernel']
    assert ck.load_variable(p1, 'transformer/encoder/encoder_fc/kernel:0').shape == (6, 4)
    assert ck.resolve(str(tmp_path)) == p1                     # directory -> newest checkpoint
    b = VariableStore(specs, torch.device('cpu'), ops.F32, seed=9)
    assert ck.restore(str(tmp_path), b) == 100 and b.adam_t == 7
    for arena in ('params', 'm', 'v'):               # by variable: the alignment padding between variables is not saved
        ea, eb = a.export(getattr(a, arena)), b.export(getattr(b, arena))
        assert all(np.array_equal(ea[k], eb[k]) for k in ea)
    # a model built under another name loads it through an assignment map
    c = VariableStore([(n.replace('transformer/', 'student/'), s, k) for n, s, k in specs], torch.device('cpu'), ops.F32, seed=1)
    ck.restore(p1, c, assignment_map={'transformer/': 'student/'})
    assert all(np.array_equal(v, c.export()[k.replace('transformer/', 'student/')]) for k, v in a.export().items())
    with pytest.raises(ValueError):
        ck.restore(p1, c)                                      # without the map the names do not match
    a.params *= 3.0
    p2 = ck.save_checkpoint(str(tmp_path / 'model-200'), a, 200, fmt=fmt)
    assert ck.resolve(str(tmp_path)) == p2
    if fmt == 'tf':
        assert open(tmp_path / 'checkpoint').read().split('\n')[0] == 'model_checkpoint_path: "model-200"'
    avg = ck.average_checkpoints([p1, p2, str(tmp_path / 'missing')], str(tmp_path / 'averaged'), fmt=fmt)
    d = VariableStore(specs, torch.device('cpu'), ops.F32, seed=5)
    assert ck.restore(avg, d) == 0
    assert all(np.allclose(v, 2.0 * b.export()[k]) for k, v in d.export().items())


@pytest.mark.parametrize('t_steps,b1,b2,want', [(8, None, None, 8), (1200, 0.0, None, 1200), (1200, 0.0, 'exact', 1200), (50000, 0.0, 0.0, 50000),
                                               (300, 'exact', None, 300)])
def test_restore_of_a_bundle_without_adam_t_recovers_the_step_from_the_beta_powers(tmp_path, t_steps, b1, b2, want):
    """(The bundle is written by this repository's own writer with the variables TensorFlow's Saver would store - no
    TensorFlow-written file exists here.)  A TF checkpoint has no adam_t, only beta1_power = 0.9^(t+1) and beta2_power = 0.98^(t+1) in fp32; beta1_power is
    exactly 0.0 after ~1000 steps (log(0) used to raise OverflowError in restore), beta2_power after ~5000."""
    import torch
    from neurst_amd import ops
    from neurst_amd.st.models.graph import VariableStore
    from neurst_amd.st.tools.tftools import checkpointTools as ck
    from neurst_amd.st.tools.tftools import tensorBundle as TB
    specs = [('transformer/encoder/encoder_fc/kernel', (6, 4), 'glorot')]
    a = VariableStore(specs, torch.device('cpu'), ops.F32, seed=3)
    t = {}
    for k, v in a.export().items():
        t[k] = v
        t[k + '/Adam'] = np.full_like(v, 0.25)
        t[k + '/Adam_1'] = np.full_like(v, 0.5)
    t['global_step'] = np.int64(t_steps)
    t['beta1_power'] = np.float32(0.9) ** np.float32(t_steps + 1) if b1 in (None, 'exact') else np.float32(b1)
    if b2 is not None:
        t['beta2_power'] = np.float32(0.98) ** np.float32(t_steps + 1) if b2 == 'exact' else np.float32(b2)
    prefix = str(tmp_path / ('model-%d' % t_steps))
    TB.write_bundle(prefix, t)
    b = VariableStore(specs, torch.device('cpu'), ops.F32, seed=9)
    assert ck.restore(prefix, b) == t_steps
    assert b.adam_t == want, b.adam_t
    assert np.array_equal(b.export(b.m)['transformer/encoder/encoder_fc/kernel'], t['transformer/encoder/encoder_fc/kernel/Adam'])


def test_tensor_bundle_container(tmp_path):
    """tensorBundle.py: the LevelDB-format index (prefix-compressed keys, several blocks, footer magic), block and
    tensor checksums, a snappy-compressed block, dtype / shape coverage and a TensorFlow-written state file."""
    import struct
    from neurst_amd.st.tools.tftools import checkpointTools as ck
    from neurst_amd.st.tools.tftools import tensorBundle as TB
    rng = np.random.default_rng(0)
    t = {'transformer/encoder/block_%d/ff_hidden/dense/kernel' % i: rng.standard_normal((7, 5)).astype(np.float32) for i in range(300)}
    t.update({'global_step': np.int64(1234), 'beta1_power': np.float32(0.9 ** 8), 'x/empty': np.zeros((0, 3), np.float32),
              'x/i32': np.arange(6, dtype=np.int32).reshape(2, 3), 'x/f64': rng.standard_normal(3), 'x/bool': np.array([True, False])})
    prefix = str(tmp_path / 'model-5')
    TB.write_bundle(prefix, t)
    index = open(prefix + '.index', 'rb').read()
    assert struct.unpack('<Q', index[-8:])[0] == 0xdb4775248b80fb57 and len(index) > 3 * 4096      # several data blocks
    assert len(index) < sum(len(k) + 30 for k in t)                                                # keys are prefix-compressed
    r = TB.BundleReader(prefix)
    assert sorted(r.entries) == sorted(t)
    for k, v in t.items():
        g = r.get_tensor(k)
        assert g.dtype == np.asarray(v).dtype and g.shape == np.asarray(v).shape and np.array_equal(g, v), k
    assert [k for k, _ in TB.read_table(prefix + '.index')] == sorted([b''] + [k.encode() for k in t])
    # through the checkpoint interface, with a TF-style state file naming it
    (tmp_path / 'checkpoint').write_text('model_checkpoint_path: "model-5"\nall_model_checkpoint_paths: "model-4"\nall_model_checkpoint_paths: "model-5"\n')
    assert ck.resolve(str(tmp_path)) == prefix
    assert int(ck.load_variable(str(tmp_path), 'global_step')) == 1234
    assert dict(ck.list_variables(prefix))['x/i32'] == [2, 3]
    # corruption is reported: a flipped bit in the data file (tensor crc), in the index (block crc), a bad magic number
    data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    data[10] ^= 1
    open(str(tmp_path / 'bad.data-00000-of-00001'), 'wb').write(bytes(data))
    open(str(tmp_path / 'bad.index'), 'wb').write(index)
    bad = TB.BundleReader(str(tmp_path / 'bad'))
    first = [k for k, e in bad.entries.items() if e['offset'] <= 10 < e['offset'] + e['size']][0]
    with pytest.raises(ValueError, match='checksum'):
        bad.get_tensor(first)
    flipped = bytearray(index)
    flipped[100] ^= 1
    open(str(tmp_path / 'bad2.index'), 'wb').write(bytes(flipped))
    with pytest.raises(ValueError, match='checksum'):
        TB.read_table(str(tmp_path / 'bad2.index'))
    open(str(tmp_path / 'bad3.index'), 'wb').write(index[:-1] + b'\x00')
    with pytest.raises(ValueError, match='magic'):
        TB.read_table(str(tmp_path / 'bad3.index'))
    # snappy: literal 'abcd' + copy(offset 4, length 7); a literal longer than 60 bytes; an overlapping copy of one byte
    assert TB._snappy_decompress(bytes([11, 0x0c]) + b'abcd' + bytes([((7 - 4) << 2) | 1, 4])) == b'abcdabcdabc'
    long = bytes(range(70))
    assert TB._snappy_decompress(bytes([70, 60 << 2, 69]) + long) == long
    assert TB._snappy_decompress(bytes([9, 0x00]) + b'z' + bytes([((8 - 1) << 2) | 2, 1, 0])) == b'z' * 9
    with pytest.raises(ValueError):
        TB._snappy_decompress(bytes([5, 0x00]) + b'z')
    # a hand-built table whose only data block is snappy-compressed (type byte 1) reads like the plain one
    blk = TB._BlockBuilder()
    blk.add(b'', b'\x08\x01')
    blk.add(b'aa', b'v1')
    blk.add(b'ab', b'v2')
    raw = blk.finish()
    comp = bytes([len(raw), (len(raw) - 1) << 2]) + raw        # one literal element
    body = comp + b'\x01' + struct.pack('<I', TB._crc(comp + b'\x01'))
    idx = TB._BlockBuilder(1)
    idx.add(b'ab', TB._put_varint(0) + TB._put_varint(len(comp)))
    metab, idxb = TB._BlockBuilder().finish(), idx.finish()
    off_meta = len(body)
    body += metab + b'\x00' + struct.pack('<I', TB._crc(metab + b'\x00'))
    off_idx = len(body)
    body += idxb + b'\x00' + struct.pack('<I', TB._crc(idxb + b'\x00'))
    footer = TB._put_varint(off_meta) + TB._put_varint(len(metab)) + TB._put_varint(off_idx) + TB._put_varint(len(idxb))
    open(str(tmp_path / 'snappy.index'), 'wb').write(body + footer + bytes(40 - len(footer)) + struct.pack('<Q', TB.MAGIC))
    assert TB.read_table(str(tmp_path / 'snappy.index')) == [(b'', b'\x08\x01'), (b'aa', b'v1'), (b'ab', b'v2')]


def _bundle_tensors(rng):
    t = {'transformer/encoder/block_%d/ff_hidden/dense/kernel' % i: rng.standard_normal((7, 5)).astype(np.float32) for i in range(120)}
    t.update({'transformer/encoder/block_%d/layer_norm/layer_norm_scale' % i: rng.standard_normal(6).astype(np.float32) for i in range(40)})
    t.update({'global_step': np.int64(4321), 'beta1_power': np.float32(0.9 ** 8), 'x/empty': np.zeros((0, 3), np.float32),
              'x/i32': np.arange(6, dtype=np.int32).reshape(2, 3), 'x/f64': rng.standard_normal(3), 'x/bool': np.array([True, False]),
              'x/u8': np.arange(5, dtype=np.uint8), 'x/f16': rng.standard_normal((2, 2)).astype(np.float16)})
    return t


@pytest.mark.parametrize('block_size,restart,compress', [(262144, 16, False), (512, 16, False), (300, 3, False), (4096, 16, True), (700, 1, True)])
def test_bundle_reader_against_the_independent_oracle_writer(tmp_path, block_size, restart, compress):
    """f1, reader side: tensorBundle.BundleReader / checkpointTools read files written by oracle/bundle.py - a restatement
    of TensorFlow's TableBuilder + BundleWriter by other means (google.protobuf messages, table-driven CRC, shortest-separator
    index keys) - in TF's own geometry (one 256 KiB block, restart interval 16) and in geometries that force many blocks,
    short restart runs and snappy-compressed blocks; values, dtypes and shapes must come back exactly, bfloat16 widened."""
    from oracle import bundle as OB
    from neurst_amd.st.tools.tftools import checkpointTools as ck
    from neurst_amd.st.tools.tftools import tensorBundle as TB
    rng = np.random.default_rng(block_size + restart)
    t = _bundle_tensors(rng)
    bf = rng.standard_normal((3, 4)).astype(np.float32)
    bf = (bf.view(np.uint32) & 0xFFFF0000).view(np.float32)
    prefix = str(tmp_path / 'model-7')
    OB.write_bundle(prefix, dict(t, **{'x/bf16': (bf, 'bfloat16')}), block_size=block_size, restart_interval=restart, compress=compress)
    tab = OB.Table(prefix + '.index')
    if block_size < 1000:
        assert tab.num_blocks() > 5
    if compress:                                                 # at least one block really is stored compressed
        kinds = []
        for _, h in tab.index:
            off, p = OB.unvarint(h, 0)
            size, _ = OB.unvarint(h, p)
            kinds.append(tab.buf[off + size])
        assert 1 in kinds
    r = TB.BundleReader(prefix)
    assert sorted(r.entries) == sorted(list(t) + ['x/bf16'])
    for k, v in t.items():
        g = r.get_tensor(k)
        assert g.dtype == np.asarray(v).dtype and g.shape == np.asarray(v).shape and np.array_equal(g, v), k
    assert np.array_equal(r.get_tensor('x/bf16'), bf)
    assert int(ck.load_variable(prefix, 'global_step')) == 4321
    assert dict(ck.list_variables(prefix))['x/i32'] == [2, 3]


def test_bundle_writer_against_the_independent_oracle_reader(tmp_path):
    """f1, writer side: what tensorBundle.write_bundle / checkpointTools.save_checkpoint(fmt='tf') write is opened by the
    oracle's reader the way TensorFlow's BundleReader does it - footer, index-block Seek, restart-array binary search,
    every block and tensor CRC recomputed from the polynomial, messages parsed by google.protobuf - tensor by tensor."""
    import torch
    from oracle import bundle as OB
    from neurst_amd import ops
    from neurst_amd.st.models.graph import VariableStore
    from neurst_amd.st.tools.tftools import checkpointTools as ck
    from neurst_amd.st.tools.tftools import tensorBundle as TB
    rng = np.random.default_rng(3)
    t = _bundle_tensors(rng)
    prefix = str(tmp_path / 'model-9')
    TB.write_bundle(prefix, t)
    r = OB.Reader(prefix)
    assert r.header.num_shards == 1 and r.header.endianness == 0 and r.header.version.producer == 1
    assert r.table.num_blocks() > 1                              # the product writes 4 KiB blocks: lookups cross the index block
    assert sorted(r.names()) == sorted(t)
    for k, v in t.items():
        g = r.tensor(k)
        assert g.dtype == np.asarray(v).dtype and g.shape == np.asarray(v).shape and np.array_equal(g, v), k
        e = r.entry(k)
        assert e.shard_id == 0 and e.size == np.asarray(v).nbytes and e.HasField('shape')
    assert r.table.get(b'no/such/tensor') is None and r.table.get(b'zzzz') is None
    # a model checkpoint under the TF names, through the checkpoint interface
    specs = [('transformer/embedding', (11, 4), 'glorot'), ('transformer/encoder/encoder_fc/kernel', (6, 4), 'glorot'),
             ('transformer/encoder/block_0/layer_norm/layer_norm_scale', (4,), 'ones')]
    a = VariableStore(specs, torch.device('cpu'), ops.F32, seed=3)
    a.m += 0.25
    a.v += 0.5
    a.adam_t = 17
    p = ck.save_checkpoint(str(tmp_path / 'model-17'), a, 17, fmt='tf')
    r = OB.Reader(p)
    P, M, V = a.export(), a.export(a.m), a.export(a.v)
    for k in P:
        assert np.array_equal(r.tensor(k), P[k]) and np.array_equal(r.tensor(k + '/Adam'), M[k]) and np.array_equal(r.tensor(k + '/Adam_1'), V[k]), k
    assert int(r.tensor('global_step')) == 17 and r.tensor('global_step').shape == ()
    assert abs(float(r.tensor('beta1_power')) - 0.9 ** 18) < 1e-6 and abs(float(r.tensor('beta2_power')) - 0.98 ** 18) < 1e-6


def test_oracle_bundle_codecs_known_answers():
    """pins of the oracle's own pieces: CRC-32C (RFC 3720 B.4), the mask constant, snappy elements from the format
    description, separator / successor keys as table_builder.cc documents them, and the two snappy decoders on each
    other's input."""
    from oracle import bundle as OB
    from neurst_amd.st.tools.tftools import tensorBundle as TB
    assert OB.crc32c(bytes(32)) == 0x8A9136AA and OB.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert OB.crc32c(bytes(range(32))) == 0x46DD794E and OB.crc32c(b'123456789') == 0xE3069283
    assert OB.crc32c(b'6789', OB.crc32c(b'12345')) == 0xE3069283                      # Extend
    for blob in (b'', b'abc', bytes(range(200))):
        assert OB.mask(OB.crc32c(blob)) == TB._crc(blob)
    assert OB.snappy_decompress(bytes([11, 0x0c]) + b'abcd' + bytes([((7 - 4) << 2) | 1, 4])) == b'abcdabcdabc'
    rng = np.random.default_rng(0)
    for blob in (b'', b'a', b'abcd' * 500, bytes(rng.integers(0, 4, 5000, dtype=np.uint8)), bytes(rng.integers(0, 256, 3000, dtype=np.uint8)),
                 b'transformer/encoder/block_0/' * 40 + bytes(300)):
        z = OB.snappy_compress(blob)
        assert OB.snappy_decompress(z) == blob and TB._snappy_decompress(z) == blob
    assert len(OB.snappy_compress(b'abcd' * 500)) < 120
    assert OB.shortest_separator(b'abc1xyz', b'abc5') == b'abc2' and OB.shortest_separator(b'abc', b'abcd') == b'abc'
    assert OB.shortest_separator(b'ab\xffz', b'ac') == b'ab\xffz' and OB.short_successor(b'\xff\xffa9') == b'\xff\xffb'


def test_avg_checkpoints_cli(tmp_path):
    """st/tools/avg_checkpoints.py flags: --checkpoints / --prefix, and --num_last_checkpoints over the state file"""
    from neurst_amd.st.tools import avg_checkpoints as avg
    from neurst_amd.st.tools.tftools import checkpointTools as ck
    from neurst_amd.st.tools.tftools import tensorBundle as TB
    for step, scale in ((100, 1.0), (200, 3.0), (300, 5.0)):
        TB.write_bundle(str(tmp_path / ('model-%d' % step)), {'w': np.full((2, 3), scale, np.float32), 'global_step': np.int64(step)})
        ck._note_in_state_file(str(tmp_path / ('model-%d' % step)))
    out = avg.main(['--checkpoints', 'model-100,model-200,model-999', '--prefix', str(tmp_path) + '/', '--output_path', str(tmp_path / 'avg_a')])
    assert os.path.exists(out + '.index') and np.array_equal(ck.load_variable(out, 'w'), np.full((2, 3), 2.0, np.float32))
    assert int(ck.load_variable(out, 'global_step')) == 0
    out = avg.main(['--num_last_checkpoints', '2', '--prefix', str(tmp_path) + '/', '--output_path', str(tmp_path / 'avg_b'), '--format', 'npz'])
    assert out.endswith('.npz') and np.array_equal(ck.load_variable(out, 'w'), np.full((2, 3), 4.0, np.float32))
    with pytest.raises(ValueError):
        avg.main(['--checkpoints', 'nope', '--prefix', str(tmp_path) + '/'])


def test_run_lut_helpers(tmp_path):
    """st/bin/run_lut.py host helpers: vocabulary file -> ids (arguments.py:92, first column, line number = id) and the
    rank sharding of the deterministic batch stream (rank r of N takes batches r, r + N, ...)"""
    from neurst_amd.st.bin import run_lut
    (tmp_path / 'vocab.txt').write_text('<pad>\n<unk> 7\n<sos>\n<eos>\nhello 12\n\nworld\n<blk>\n', encoding='utf-8')
    t2i, i2t = run_lut.load_vocab(str(tmp_path / 'vocab.txt'))
    assert t2i == {'<pad>': 0, '<unk>': 1, '<sos>': 2, '<eos>': 3, 'hello': 4, 'world': 5, '<blk>': 6} and i2t[5] == 'world'
    stream = list(range(23))
    shards = [list(run_lut._Shard(stream, r, 4)) for r in range(4)]
    assert shards[0] == [0, 4, 8, 12, 16] and shards[3] == [3, 7, 11, 15, 19]     # whole rounds only: every rank takes the same number of steps
    assert sorted(sum(shards, [])) == list(range(20))


EARLY_STOP_WORKER = r"""
import os, sys, faulthandler
faulthandler.dump_traceback_later(90, exit=True)
sys.path.insert(0, %r)
import torch, torch.distributed as dist
rank = int(sys.argv[1])
os.environ.update(RANK=str(rank), WORLD_SIZE='2')
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + sys.argv[2], rank=rank, world_size=2)
from neurst_amd.st.bin import run_lut
from neurst_amd.st.layers.arguments import AttrDict

class Step:
    value = 0
class FakeModel:
    # the gradient collective of the real step: a rank that keeps training after the others stopped blocks in here
    def __init__(self):
        self.global_step, self.last, self.store = Step(), {'lr': 1e-3}, None
    def train_step(self, batch):
        t = torch.ones(4)
        dist.all_reduce(t)
        assert float(t[0]) == 2.0
        self.global_step.value += 1
        return 1.0 / self.global_step.value, (2, 10, 4), 0.1, 0.2, 0.3
class Reader:
    def __init__(self, *a, **k): pass
    def __iter__(self): return iter(range(10 ** 9))
class Pre:
    def __init__(self, it, depth=4): self.it = iter(it)
    def __next__(self): return next(self.it)
    def __iter__(self): return self
    def close(self): pass
class Summ:
    def __init__(self, *a): pass
    def summary_scalar(self, *a): pass
dev_calls = []
def fake_dev_test(args, step, *a, **k):
    assert rank == 0, 'only rank 0 scores the dev set'
    dev_calls.append(step)
    return 0.5, 0.5 + 0.01 * len(dev_calls), 10.0 - len(dev_calls)     # never better than the first: bad_count grows
run_lut.build_models = lambda args: (FakeModel(), FakeModel())
run_lut.TFReader, run_lut.Prefetcher, run_lut.Summary, run_lut.dev_test = Reader, Pre, Summ, fake_dev_test
run_lut.checkpointTools.save_checkpoint = lambda path, *a, **k: path
args = AttrDict({'dir_checkpoint': sys.argv[3], 'dir_log': sys.argv[3], 'dirs': {'train': {'tfdata': 'x'}, 'dev': {'tfdata': 'y'}},
                 'metrics': 'BLEU', 'early_stop': 2, 'dev_step': 3, 'num_epochs': 1000, 'data': {'unit': 'word'}, 'dim_output': 11, 'eos_idx': 3})
losses = run_lut.train(args)
# dev passes at step 0, 3, 6, 9, ...: the first saves, the next three are worse -> bad_count 3 > early_stop 2 at step 9
assert len(losses) == 10, len(losses)
if rank == 0:
    assert dev_calls == [0, 3, 6, 9], dev_calls
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'stopped after', len(losses), 'steps')
"""


def test_early_stop_reaches_every_rank(tmp_path):
    """st/bin/run_lut.py:420-422 with one process per GPU: rank 0 alone scores the dev set, but its early-stop decision
    must end the loop on EVERY rank - a rank that trained on would block in the next gradient all-reduce for good."""
    script = tmp_path / 'worker.py'
    script.write_text(EARLY_STOP_WORKER % ROOT)
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=150)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all('stopped after 10 steps' in o for o in outs), outs
