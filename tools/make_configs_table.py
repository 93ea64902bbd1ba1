#!/usr/bin/env python3
"""gpurun_out/configs/*.json (tools/run_configs.sh) -> the table under profiles/<tag>_configs.txt."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
rows = []
for path in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'configs', '*.json'))):
    lines = [l for l in open(path).read().splitlines() if l.startswith('{')]
    if not lines:
        continue
    d = json.loads(lines[-1])
    r = d.get('roofline', {})
    gf = d['config']['train_gflop_per_step_per_gpu']
    us = 1e3 * d['ms_per_step']
    rows.append('%-24s %-5s %9d %9.1f %11.3f %12.1f %7.3f  %-42s %8.1f %6.3f %9s' % (
        os.path.basename(path)[:-5], d['dtype'], d['value'], us, gf, gf / us * 1e3, r.get('step_frac', 0.0),
        (r.get('kernel', '') + ' ' + r.get('gemm', '').split(' ')[0])[:42], r.get('achieved', 0.0), r.get('frac', 0.0),
        int(d['cpu_baseline']['value']) if 'cpu_baseline' in d else '-'))
head = """# One MI355X, `bash tools/run_configs.sh` (bench.py --config ... --batch ... --steps 100): every BASELINE.json config at its
# per-GPU batch, fp32 (the parity path) and, where the fused step applies, the opt-in bf16 path.  lists/s = whole step
# (device feed + forward + loss + backward + clip/Adagrad + in-step metrics); step TFLOP/s = SURVEY 8d GEMM flops of the
# step / step time, frac against the fp32 MFMA peak 157.3 in both modes; gemm = the dominant GEMM launch of that run with its
# own algorithmic TFLOP/s, frac against the peak of its dtype (bf16: 2500); cpu = the op-for-op CPU restatement of the
# reference on the box's host cores (bounded sample).  c1 runs on the reference's toy libsvm data (27 queries, lists of
# 9), the others on synthetic data.
%-24s %-5s %9s %9s %11s %12s %7s  %-42s %8s %6s %9s""" % ('config', 'dtype', 'lists/s', 'us/step', 'GFLOP/step', 'step TFLOP/s', 'frac',
                                                         'dominant GEMM launch', 'TFLOP/s', 'frac', 'cpu l/s')
out = head + '\n' + '\n'.join(rows) + '\n'
open(os.path.join(ROOT, 'profiles', tag + '_configs.txt'), 'w').write(out)
print(out)
