#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid, workgroup): launches, mean/min duration, gaps.

    python tools/trace_summary.py gpurun_out/prof/<host>/<pid>_kernel_trace.csv [--last N]

The per-grid split separates the layers of one templated kernel (fwd_layer_kernel is launched once per
hidden layer with a different grid), which the --stats summary folds together.
"""
import csv
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r'\(.*$', '', name)
    name = re.sub(r'^void ', '', name)
    return name[-60:]


def main():
    path = sys.argv[1]
    last = int(sys.argv[sys.argv.index('--last') + 1]) if '--last' in sys.argv else 0
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    if last:
        rows = rows[-last:]
    groups = OrderedDict()
    prev_end = None
    for r in rows:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        key = (short(r['Kernel_Name']), '%sx%sx%s' % (r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z']),
               r['Workgroup_Size_X'], r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'])
        g = groups.setdefault(key, {'n': 0, 'sum': 0, 'min': 1 << 60, 'gap': 0})
        g['n'] += 1
        g['sum'] += e - s
        g['min'] = min(g['min'], e - s)
        if prev_end is not None:
            g['gap'] += max(0, s - prev_end)
        prev_end = e
    print('%-60s %-16s %5s %5s %5s %6s %5s %6s %9s %9s %9s' % (
        'kernel', 'grid(threads)', 'wg', 'vgpr', 'agpr', 'lds', 'scr', 'n', 'mean_us', 'min_us', 'gap_us'))
    tot = 0.0
    for key, g in groups.items():
        print('%-60s %-16s %5s %5s %5s %6s %5s %6d %9.2f %9.2f %9.2f' % (
            key + (g['n'], g['sum'] / g['n'] / 1e3, g['min'] / 1e3, g['gap'] / g['n'] / 1e3)))
        tot += g['sum']
    print('total kernel time %.1f us over %d dispatches' % (tot / 1e3, len(rows)))


if __name__ == '__main__':
    main()
