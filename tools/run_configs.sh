#!/bin/bash
# One bench.py line per BASELINE.json config on ONE GPU (per-GPU batch as the config states); output under gpurun_out/configs/
set -e
out=gpurun_out/configs
mkdir -p $out
run() { name=$1; shift; timeout -k 10 200 python bench.py --steps 100 --warmup 10 --cpu_seconds 4 "$@" > $out/$name.json 2> $out/$name.err; tail -1 $out/$name.json | cut -c1-400; }
run c1_toy_ipw_b256 --config toy --batch 256
run c2_mslr10k_b256 --config mslr10k --batch 256
run c3_yahoo_b256 --config yahoo --batch 256
run c4_istella_b256 --config istella --batch 256
run c5_mslr30k_b4096 --config mslr30k --batch 4096
run c3_yahoo_b4096 --config yahoo --batch 4096
run c2_mslr10k_b256_bf16 --config mslr10k --batch 256 --dtype bf16
run c3_yahoo_b256_bf16 --config yahoo --batch 256 --dtype bf16
run c4_istella_b256_bf16 --config istella --batch 256 --dtype bf16
run c5_mslr30k_b4096_bf16 --config mslr30k --batch 4096 --dtype bf16
