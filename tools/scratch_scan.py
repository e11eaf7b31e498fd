#!/usr/bin/env python
"""Scratch (register spill) traffic of every kernel, from the compiler's assembly: a reload is followed by s_waitcnt vmcnt(0),
which also waits for every store / LDS-DMA piece the wave issued before it (one counter, in order), so a reload between two
stores turns them into dependent HBM round trips (found in the fused attention backward: 36 k cycles per pass of dK / dV
stores).  Usage: python tools/scratch_scan.py [file.hip ...]   (default: every .hip of neurst_amd/csrc)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'neurst_amd', 'csrc')


def demangle(n):
    try:
        return subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def scan(path):
    with tempfile.NamedTemporaryFile(suffix='.s') as tmp:
        subprocess.run(['hipcc', '-O3', '--offload-arch=gfx950', '-std=c++17', '-Wno-unused-value', '-S', '--cuda-device-only', path, '-o', tmp.name],
                       check=True, stderr=subprocess.DEVNULL, cwd=SRC)
        text = open(tmp.name).read()
    out = []
    for f in re.split(r'\n(?=_Z\w+:\s)', text):
        name = f.split(':')[0]
        if not name.startswith('_Z'):
            continue
        lines = f.split('\n')
        loads = [i for i, l in enumerate(lines) if 'scratch_load' in l]
        stores = sum('scratch_store' in l for l in lines)
        if not loads and not stores:
            continue
        labels = [(i, l) for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)]
        in_loop = 0
        for k, (i, l) in enumerate(labels):
            j = labels[k + 1][0] if k + 1 < len(labels) else len(lines)
            if 'Loop' in l:
                in_loop += sum('scratch_load' in x for x in lines[i:j])
        # reloads with a store / DMA piece in the ten instructions before them: the serialising kind
        near = sum(1 for i in loads if any(('global_store' in x or 'buffer_store' in x or ' lds' in x or 'global_atomic' in x) for x in lines[max(0, i - 10):i]))
        out.append((os.path.basename(path), len(loads), stores, in_loop, near, demangle(name)[:120]))
    return out


if __name__ == '__main__':
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(SRC, '*.hip')))
    print('%-16s %6s %6s %8s %10s  kernel' % ('file', 'loads', 'stores', 'in-loops', 'after-st'))
    for p in files:
        for r in scan(os.path.abspath(p)):
            print('%-16s %6d %6d %8d %10d  %s' % r)
