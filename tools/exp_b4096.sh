#!/bin/bash
for pf in 1 0; do
  echo "== prefetch=$pf"
  timeout -k 10 120 python bench.py --config mslr30k --batch 4096 --steps 50 --warmup 5 --no_cpu_baseline --prefetch $pf 2> gpurun_out/b4096_pf$pf.err | cut -c1-220
  grep -E "wgrad|finalize|assemble|sum of" gpurun_out/b4096_pf$pf.err
done
timeout -k 10 120 python bench.py --steps 100 --no_cpu_baseline --no_profile 2>/dev/null | cut -c1-220
timeout -k 10 120 python bench.py --config yahoo --steps 100 --no_cpu_baseline --no_profile 2>/dev/null | cut -c1-220
