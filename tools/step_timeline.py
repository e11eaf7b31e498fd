#!/usr/bin/env python
"""Timeline of one train step from a rocprofv3 --kernel-trace CSV (all streams; steps are delimited by adam_kernel):
   python tools/step_timeline.py <kernel_trace.csv> [summary|timeline]
summary: time per kernel name and stream per step (mean over the last three steps); timeline: every launch of the last step."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
mode = sys.argv[2] if len(sys.argv) > 2 else 'summary'
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:70], r['Stream_Id'], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])))
            for r in rows)
adam = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
if mode == 'summary':
    a, b = adam[-4], adam[-1]
    d = collections.defaultdict(lambda: [0, 0])
    for s, e, n, st, g in ev[a + 1:b + 1]:
        d[(n, st)][0] += e - s
        d[(n, st)][1] += 1
    tot = 0
    for k, v in sorted(d.items(), key=lambda x: -x[1][0]):
        print('%-72s st%s %7.1f us/step x%5.1f avg %6.1f' % (k[0], k[1], v[0] / 3e3, v[1] / 3, v[0] / v[1] / 1e3))
        tot += v[0]
    print('sum of kernel durations %.3f ms/step, wall %.3f ms/step' % (tot / 3e6, (ev[b][1] - ev[a][1]) / 3e6))
else:
    a, b = adam[-2], adam[-1]
    t0 = ev[a][1]
    for s, e, n, st, g in ev[a + 1:b + 1]:
        print('%8.1f %8.1f %7.1f st%s %s wg=%d' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, st, n, g))
