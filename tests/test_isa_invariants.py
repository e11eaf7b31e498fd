"""Static checks on the compiled code of kernels whose correctness rests on HAND-COUNTED `s_waitcnt vmcnt(n)`.

csrc/layernorm.hip: ln_fwd_line / ln_bwd_line request their rows by inline-assembly loads and wait with counts that assume
(a) the compiler issues no other vector-memory operation inside the row loop (no scratch traffic, one store per chunk) and
(b) it keeps its stores and the assembly loads in program order.  Neither is visible at run time until a wait passes too early,
so the ISA of every instantiation is checked here (hipcc -S for gfx950: no GPU needed)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'neurst_amd', 'csrc')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


@pytest.fixture(scope='module')
def layernorm_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not available')
    out = tmp_path_factory.mktemp('isa') / 'layernorm.s'
    subprocess.run([HIPCC, '-O3', '--offload-arch=gfx950', '-std=c++17', '-Wno-unused-value', '-S', '--cuda-device-only',
                    os.path.join(CSRC, 'layernorm.hip'), '-o', str(out)], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return open(out).read()


def kernels(isa, prefix):
    """name -> (body text, private segment size) of every kernel whose mangled name starts with `prefix`"""
    found = {}
    for m in re.finditer(r'^(%s\w*):\s*;.*?$' % re.escape(prefix), isa, re.M):
        name = m.group(1)
        end = isa.index('s_endpgm', m.end())
        seg = re.search(r'\.set %s\.private_seg_size, (\d+)' % re.escape(name), isa)
        found[name] = (isa[m.end():end], int(seg.group(1)))
    return found


def vm_ops(text):
    """the vector-memory operations and vmcnt waits of a piece of ISA, in program order: 'L' load, 'S' store, ('W', n), 'X' other"""
    ops = []
    for line in text.splitlines():
        t = line.strip()
        if t.startswith(('global_load', 'buffer_load', 'flat_load')):
            ops.append('L')
        elif t.startswith(('global_store', 'buffer_store', 'flat_store')):
            ops.append('S')
        elif t.startswith(('scratch_', 'global_atomic', 'buffer_atomic', 'flat_atomic')):
            ops.append('X')
        else:
            m = re.match(r's_waitcnt\b.*vmcnt\((\d+)\)', t)
            if m:
                ops.append(('W', int(m.group(1))))
    return ops


@pytest.mark.parametrize('which', ['fwd', 'bwd'])
def test_layernorm_line_kernels_issue_exactly_the_operations_their_waits_count(layernorm_isa, which):
    ks = kernels(layernorm_isa, '_Z11ln_%s_lineI' % which)
    assert len(ks) == 6, sorted(ks)          # (VEC, NCH) in (4, 1), (8, 1), (4, 3) with and without dropout
    for name, (body, scratch) in ks.items():
        vec, nch, drop = (int(x) for x in re.match(r'_Z11ln_..._lineILi(\d)ELi(\d)ELb([01])E', name).groups())
        assert scratch == 0, (name, 'scratch traffic goes through the vector-memory counter')
        nl, ns = (2 * nch, 2 * nch + 2) if which == 'fwd' else (2 * nch + 2, nch * (2 if drop else 1))
        # the row loop: its header block and every block the assembler lists as part of it ("in Loop: Header=..."), in program
        # order from the header; blocks of the loop placed in front of the header (the latch) must not touch memory
        lines = body.splitlines()
        hdr = [i for i, l in enumerate(lines) if re.match(r'\.LBB\d+_\d+:\s*; =>This Inner Loop Header', l)]
        assert len(hdr) == 1, (name, 'one row loop')
        tag = 'Header=BB' + re.match(r'\.LBB(\d+_\d+):', lines[hdr[0]]).group(1)
        loop, i = [], hdr[0] + 1
        while i < len(lines) and not (re.match(r'\.LBB\d+_\d+:', lines[i]) and tag not in lines[i]):
            loop.append(lines[i])
            i += 1
        front = []
        for j, l in enumerate(lines[:hdr[0]]):
            if re.match(r'\.LBB\d+_\d+:', l) and tag in l:
                k = j + 1
                while k < hdr[0] and not re.match(r'\.LBB\d+_\d+:', lines[k]):
                    front.append(lines[k])
                    k += 1
        assert vm_ops('\n'.join(front)) == [], (name, 'the loop latch issues memory operations')
        ops = vm_ops('\n'.join(loop))
        want = (['L'] * nl + [('W', ns + nl)] + ['S'] * ns) * 2
        assert ops == want, (name, ops)
