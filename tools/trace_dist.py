"""Per-kernel duration distribution and inter-kernel gaps from a rocprofv3 kernel_trace.csv (runs on the GPU box)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = defaultdict(list)
gap = defaultdict(list)
prev = None
for r in rows:
    n = r['Kernel_Name'].split('(')[0][-40:]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d[n].append((e - s) / 1e3)
    if prev is not None:
        gap[n].append((s - prev) / 1e3)
    prev = e
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 50:
        continue
    v2 = sorted(v[len(v) // 3:])  # past warm-up
    g = sorted(gap[n][len(gap[n]) // 3:])
    q = lambda a, f: a[min(len(a) - 1, int(f * len(a)))]
    print('%-42s n=%4d dur min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f | gap-before p50 %.1f p90 %.1f' % (
        n, len(v), v2[0], q(v2, .1), q(v2, .5), q(v2, .9), v2[-1], q(g, .5), q(g, .9)))
