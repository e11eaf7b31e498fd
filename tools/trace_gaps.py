#!/usr/bin/env python
"""Idle time of the GPU inside a train step, from a rocprofv3 --kernel-trace CSV (kernel_trace.csv):
union of kernel intervals over all streams vs wall time of the last steps, and the largest gaps with the
kernels on either side.  Usage: python tools/trace_gaps.py <..._kernel_trace.csv> [n_last_kernels]"""
import csv
import sys


def short(n):
    return n.split('(')[0][:60]


rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows), key=lambda e: e[0])
# steps are delimited by adam_kernel launches
adam = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
assert len(adam) >= 4, 'need a few steps'
a, b = adam[-4], adam[-1]                       # three whole steps
seg = ev[a + 1:b + 1]
t0, t1 = ev[a][1], ev[b][1]
busy, gaps = 0, []
cur_s, cur_e, last = seg[0][0], seg[0][1], seg[0][2]
prev_end_name = ev[a][2]
if seg[0][0] > t0:
    gaps.append((seg[0][0] - t0, prev_end_name, seg[0][2]))
for s, e, n in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last, n))
        cur_s, cur_e, last = s, e, n
    elif e > cur_e:
        cur_e, last = e, n
busy += cur_e - cur_s
wall = t1 - t0
print('3 steps: wall %.3f ms/step, GPU busy (union over streams) %.3f ms/step, idle %.3f ms/step in %d gaps/step' % (
    wall / 3e6, busy / 3e6, (wall - busy) / 3e6, len(gaps) // 3))
ksum = sum(e - s for s, e, _ in seg)
print('sum of kernel durations %.3f ms/step (overlap between streams %.3f ms/step)' % (ksum / 3e6, (ksum - busy) / 3e6))
hist = {}
for g, p, n in gaps:
    k = (p, n)
    hist.setdefault(k, [0, 0])
    hist[k][0] += g
    hist[k][1] += 1
print('idle time by (kernel before gap -> kernel after), per step:')
for (p, n), (tot, cnt) in sorted(hist.items(), key=lambda kv: -kv[1][0])[:18]:
    print('  %7.1f us  x%-4.1f avg %5.1f us   %s -> %s' % (tot / 3e3, cnt / 3, tot / cnt / 1e3, p[:42], n[:42]))
