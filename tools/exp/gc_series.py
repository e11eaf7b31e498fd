#!/usr/bin/env python
"""the step series of step_series.py with Python's cycle collector timed (gc.callbacks) or switched off (GC=off)"""
import gc
import os
import sys
import time

_t = {}
log = []


def cb(phase, info):
    if phase == 'start':
        _t['t'] = time.perf_counter()
    else:
        log.append((info['generation'], (time.perf_counter() - _t['t']) * 1e3, info['collected']))


gc.callbacks.append(cb)
if os.environ.get('GC') == 'off':
    gc.disable()
sys.argv = [sys.argv[0]] + sys.argv[1:]
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, 'step_series.py')).read()
n0 = len(gc.get_objects())
exec(compile(src, 'step_series.py', 'exec'))
print('objects tracked before / after', n0, len(gc.get_objects()))
for gen in (0, 1, 2):
    xs = [d for g, d, c in log if g == gen]
    print('gen', gen, 'collections', len(xs), 'total ms %.1f' % sum(xs), 'max ms %.1f' % (max(xs) if xs else 0))
