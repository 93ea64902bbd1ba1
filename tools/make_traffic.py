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
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
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


if __name__ == '__main__':
    main()
