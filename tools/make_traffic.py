#!/usr/bin/env python3
"""Turn the rocprofv3 passes of tools/collect_profiles.sh into the committed evidence under profiles/:

  profiles/<tag>_kernel_stats.csv     --kernel-trace --stats summary of the default bench command
  profiles/<tag>_kernel_by_grid.txt   the same trace split per (kernel, grid)
  profiles/<tag>_traffic.json         HBM-side bytes per launch and kernel from the PMC passes

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies each 128-byte request as 64 bytes
(MI355X_MICROARCH.md, "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read"), so
the read side is doubled; WRITE_SIZE is exact.  bench.py reads <tag>_traffic.json for roofline.traffic.
"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name).replace('ultr::', '')


def per_kernel_mean(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        a = acc[short(r['Kernel_Name'])]
        a[0] += float(r['Counter_Value'])
        a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


def newest(pattern):
    """gpurun merges every call's output into gpurun_out/: take the latest pass, not an earlier one."""
    return max(glob.glob(pattern), key=os.path.getmtime)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
    src = os.path.join(ROOT, 'gpurun_out', 'prof_' + tag)
    dst = os.path.join(ROOT, 'profiles')
    os.makedirs(dst, exist_ok=True)
    stats = newest(os.path.join(src, 'stats', '*', '*_kernel_stats.csv'))
    shutil.copy(stats, os.path.join(dst, tag + '_kernel_stats.csv'))
    trace = newest(os.path.join(src, 'stats', '*', '*_kernel_trace.csv'))
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'trace_summary.py'), trace], capture_output=True, text=True)
    open(os.path.join(dst, tag + '_kernel_by_grid.txt'), 'w').write(out.stdout)
    fetch = per_kernel_mean(newest(os.path.join(src, 'fetch', '*', '*_counter_collection.csv')), 'FETCH_SIZE')
    write = per_kernel_mean(newest(os.path.join(src, 'write', '*', '*_counter_collection.csv')), 'WRITE_SIZE')
    res = {'unit': 'bytes per launch', 'correction': 'FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies 128-B requests as 64 B) + WRITE_SIZE KiB x 1024',
           'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_profile --graph 0',
           'kernels': {}}
    for k in sorted(set(fetch) | set(write)):
        if 'ultr' not in k and not any(s in k for s in ('_kernel',)):
            continue
        f, w = fetch.get(k, 0.0) * 1024 * 2, write.get(k, 0.0) * 1024
        res['kernels'][k] = {'fetch_bytes': round(f), 'write_bytes': round(w), 'hbm_bytes': round(f + w)}
    json.dump(res, open(os.path.join(dst, tag + '_traffic.json'), 'w'), indent=1, sort_keys=True)
    for k, v in res['kernels'].items():
        print('%-60s fetch %10d  write %10d' % (k[:60], v['fetch_bytes'], v['write_bytes']))
    sq = glob.glob(os.path.join(src, 'sq', '*', '*_counter_collection.csv'))
    if sq:
        path = max(sq, key=os.path.getmtime)
        names = ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY',
                 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU_MFMA_MOPS_F32', 'SQ_LDS_BANK_CONFLICT', 'GRBM_GUI_ACTIVE']
        per = {n: per_kernel_mean(path, n) for n in names}
        lines = ['# per launch, mean over the launches of the pass (rocprofv3 --pmc, tools/collect_profiles.sh pass 4)',
                 '# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs); GRBM_GUI_ACTIVE is',
                 '# summed over the 8 XCDs (MI355X_MICROARCH.md) and reads high on launches this short, so the figure is a',
                 '# lower bound; wait/active are shares of SQ_WAVE_CYCLES (quad-cycles)',
                 '%-44s %12s %12s %9s %8s %8s %8s %12s %10s' % ('kernel', 'mfma_busy', 'gui_active', 'mfma_util', 'wait', 'stall',
                                                               'active', 'mfma_mops', 'lds_confl')]
        for k in sorted(per['SQ_WAVE_CYCLES']):
            if 'ultr' not in k and '_kernel' not in k:
                continue
            g = lambda n: per[n].get(k, 0.0)
            wc = max(g('SQ_WAVE_CYCLES'), 1.0)
            util = g('SQ_VALU_MFMA_BUSY_CYCLES') / max(g('GRBM_GUI_ACTIVE') / 8.0 * 256 * 4, 1.0)
            lines.append('%-44s %12.0f %12.0f %8.1f%% %7.1f%% %7.1f%% %7.1f%% %12.0f %10.0f' % (
                k[:44], g('SQ_VALU_MFMA_BUSY_CYCLES'), g('GRBM_GUI_ACTIVE'), 100 * util, 100 * g('SQ_WAIT_ANY') / wc,
                100 * g('SQ_WAIT_INST_ANY') / wc, 100 * g('SQ_ACTIVE_INST_ANY') / wc, g('SQ_INSTS_VALU_MFMA_MOPS_F32'),
                g('SQ_LDS_BANK_CONFLICT')))
        open(os.path.join(dst, tag + '_mfma_util.txt'), 'w').write('\n'.join(lines) + '\n')
        print('\n'.join(lines))


if __name__ == '__main__':
    main()
