#!/bin/bash
# large-batch experiment: C5 (MSLR-30K shape, batch 4096) and the Yahoo shape at batch 4096, kernel table included
for cfg in mslr30k yahoo; do
  timeout -k 10 150 python bench.py --config $cfg --batch 4096 --steps 50 --warmup 5 --no_cpu_baseline 2> gpurun_out/b4096_$cfg.err | cut -c1-220
  grep -E "us/launch|sum of" gpurun_out/b4096_$cfg.err | head -8
done
