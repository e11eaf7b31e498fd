#!/usr/bin/env python
"""Turn the rocprofv3 output of one bench.py run (gpurun_out/<tag>/{stats,fetch,write}) into the committed summaries:
  profiles/<tag>_bench_lut_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (copied verbatim)
  profiles/<tag>_hbm_traffic.json             per-kernel HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> B),
                                              the gfx950 correction of MI355X_MICROARCH.md 'HBM', from two separate --pmc passes
  profiles/<tag>_mfma_util.json               per-kernel MfmaUtil (matrix-pipe busy share) from a --pmc MfmaUtil pass, when gpurun_out/<tag>/mfma exists
Usage: python tools/make_profiles.py r01"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gemm_label(name):
    """mangled / half-demangled gemm_kernel<T, TC, TA, TB_KCONTIG, FAST, SUMOP, NS> -> the label bench.py uses"""
    if 'gemm256_group_kernel' in name:
        return 'gemm256_group_kernel'
    m2 = re.match(r'_ZN2t214gemm256_kernelI(DF16b|f)Lb([01])ELb([01])E', name)
    if m2:
        lay = {('0', '1'): 'NT', ('0', '0'): 'NN', ('1', '0'): 'TN', ('1', '1'): 'TT'}[m2.groups()[1:]]
        return 'gemm256_kernel<bf16,%s,%s>' % ('bf16' if m2.group(1) == 'DF16b' else 'f32', lay)
    m2 = re.match(r'void t2::gemm256_kernel<(float|bool _Accum|__bf16), (true|false), (true|false)', name)
    if m2:
        lay = {('false', 'true'): 'NT', ('false', 'false'): 'NN', ('true', 'false'): 'TN', ('true', 'true'): 'TT'}[m2.groups()[1:]]
        return 'gemm256_kernel<bf16,%s,%s>' % ('f32' if m2.group(1) == 'float' else 'bf16', lay)
    # (the instantiations of one label differ in their last template arguments: what the epilogue knows at compile time)
    mp = re.match(r'_Z19gemm_persist_kernelI(DF16b|f)Lb([01])ELb([01])E(?:Lb[01]E)?Ev8GemmArgs', name)
    if mp:
        lay = {('0', '1'): 'NT', ('0', '0'): 'NN', ('1', '0'): 'TN'}[mp.groups()[1:]]
        return 'gemm_persist_kernel<bf16,%s,%s>' % ('bf16' if mp.group(1) == 'DF16b' else 'f32', lay)
    mp = re.match(r'void gemm_persist_kernel<(float|bool _Accum|__bf16), (true|false), (true|false)', name)
    if mp:
        lay = {('false', 'true'): 'NT', ('false', 'false'): 'NN', ('true', 'false'): 'TN'}[mp.groups()[1:]]
        return 'gemm_persist_kernel<bf16,%s,%s>' % ('f32' if mp.group(1) == 'float' else 'bf16', lay)
    mt = re.match(r'_Z16gemm_thin_kernelI(DF16b|f)Lb([01])E(?:Lb[01]E)?(?:Li\d+E)?Ev8GemmArgs', name)
    if mt:
        return 'gemm_thin_kernel<bf16,%s,%s>' % ('bf16' if mt.group(1) == 'DF16b' else 'f32', 'NT' if mt.group(2) == '1' else 'NN')
    mt = re.match(r'void gemm_thin_kernel<(float|bool _Accum|__bf16), (true|false|bool)', name)   # rocprofv3's half-demangled spellings
    if mt:
        return 'gemm_thin_kernel<bf16,%s,%s>' % ('f32' if mt.group(1) == 'float' else 'bf16', 'NN' if mt.group(2) == 'false' else 'NT')
    m = re.match(r'_Z11gemm_kernelI(DF16b|f)(DF16b|f)Lb([01])ELb([01])ELb([01])ELi(\d)ELi(\d)E(?:Li\dE)?Ev8GemmArgs', name)
    if m:
        t, tc, ta, tbk, fast, sumop, ns = m.groups()
    else:
        m = re.match(r'void gemm_kernel<bool _Accum, (true|false), (true|false), (true|false), (\d), (\d)(?:, \d)?>', name)
        if not m:
            return None
        t, tc = 'DF16b', 'f'
        ta, tbk, fast = ['1' if x == 'true' else '0' for x in m.groups()[:3]]
        sumop, ns = m.groups()[3:]
    lay = {('0', '1'): 'NT', ('0', '0'): 'NN', ('1', '0'): 'TN', ('1', '1'): 'TT'}[(ta, tbk)]
    return 'gemm_kernel<%s,%s,%s,%s,stages=%s%s>' % ('bf16' if t == 'DF16b' else 'f32', 'bf16' if tc == 'DF16b' else 'f32', lay,
                                                     'direct-to-LDS' if fast == '1' else 'register-staged', ns, ',+bias-grad' if sumop != '0' else '')


def counters(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r['Kernel_Name']].append(float(r['Counter_Value']))
    return d


def main(tag):
    src = os.path.join(ROOT, 'gpurun_out', tag)
    dst = os.path.join(ROOT, 'profiles')
    os.makedirs(dst, exist_ok=True)
    sys.path.insert(0, ROOT)
    from bench import kernel_sources_digest
    shutil.copy(os.path.join(src, 'stats', 's_kernel_stats.csv'), os.path.join(dst, tag + '_bench_lut_kernel_stats.csv'))
    f = counters(os.path.join(src, 'fetch', 'f_counter_collection.csv'))
    w = counters(os.path.join(src, 'write', 'w_counter_collection.csv'))
    out = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --serial-streams',
           'formula': 'bytes = 1000 * (2 * mean(FETCH_SIZE) + mean(WRITE_SIZE)); FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (gfx950 tallies 128-B reads at 64 B); Infinity-Cache hits are included',
           'kernel_sources_digest': kernel_sources_digest(),
           'bytes_per_launch': {}, 'raw_kernel_names': {}}
    for k in sorted(f):
        fb = sum(f[k]) / len(f[k])
        wb = sum(w[k]) / len(w[k]) if k in w else 0.0
        label = gemm_label(k) or k.split('(')[0]
        # several instantiations can share a label (tile widths of gemm256): keep the launch-weighted mean
        prev = out['raw_kernel_names'].get(label)
        val = 1000 * (2 * fb + wb)
        if prev:
            n0, n1 = prev['launches'], len(f[k])
            val = (out['bytes_per_launch'][label] * n0 + val * n1) / (n0 + n1)
        out['bytes_per_launch'][label] = round(val)
        out['raw_kernel_names'][label] = {'name': k, 'launches': len(f[k]) + (prev['launches'] if prev else 0), 'fetch_kb': round(fb, 1), 'write_kb': round(wb, 1)}
    json.dump(out, open(os.path.join(dst, tag + '_hbm_traffic.json'), 'w'), indent=1)
    mf = os.path.join(src, 'mfma', 'm_counter_collection.csv')
    if os.path.exists(mf):            # a fourth pass: --pmc MfmaUtil
        util, dur = collections.defaultdict(list), collections.defaultdict(list)
        for r in csv.DictReader(open(mf)):
            k = gemm_label(r['Kernel_Name']) or r['Kernel_Name'].split('(')[0]
            util[k].append(float(r['Counter_Value']))
            dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        kernels = {k: {'launches': len(v), 'mfma_util_mean_percent': round(sum(v) / len(v), 1), 'mfma_util_max_percent': round(max(v), 1),
                       'avg_us_under_counters': round(sum(dur[k]) / len(v), 1)}
                   for k, v in sorted(util.items(), key=lambda kv: -sum(dur[kv[0]])) if max(v) > 0}
        json.dump({'source': 'rocprofv3 --pmc MfmaUtil -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --serial-streams',
                   'formula': 'MfmaUtil = 100 * sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM): share of the cycles a kernel was on the GPU in '
                              'which a SIMD had its matrix pipe busy, at the clock the chip held during that kernel',
                   'kernels': kernels}, open(os.path.join(dst, tag + '_mfma_util.json'), 'w'), indent=1)
    lf = os.path.join(src, 'lds', 'l_counter_collection.csv')
    if os.path.exists(lf):            # a fifth pass: --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.defaultdict(int)
        for r in csv.DictReader(open(lf)):
            k = gemm_label(r['Kernel_Name']) or r['Kernel_Name'].split('(')[0]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            n[k] += r['Counter_Name'] == 'SQ_LDS_IDX_ACTIVE'
        kernels = {k: {'launches': n[k], 'lds_bank_conflict_cycles': round(v.get('SQ_LDS_BANK_CONFLICT', 0.0)),
                       'lds_active_cycles': round(v.get('SQ_LDS_IDX_ACTIVE', 0.0)),
                       'conflict_share_percent': round(100.0 * v.get('SQ_LDS_BANK_CONFLICT', 0.0) / v['SQ_LDS_IDX_ACTIVE'], 1)}
                   for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0.0)) if v.get('SQ_LDS_IDX_ACTIVE', 0.0) > 0}
        json.dump({'source': 'rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --serial-streams',
                   'formula': 'conflict_share = cycles the LDS spent on bank conflicts / cycles it was active, both summed over a kernel\'s launches',
                   'kernels': kernels}, open(os.path.join(dst, tag + '_lds_conflicts.json'), 'w'), indent=1)
    print('wrote', os.listdir(dst))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'r01')
