#!/bin/bash
# The other BASELINE.json configurations through bench.py, each WITH the CPU checker on (cpu_baseline.max_abs_logit_diff_vs_gpu):
# config 2 (ViT-B/14 batch 1), config 4's single-GPU share (ViT-g/14 bf16 batch 8), config 5 (ViT-L q8_0 / q4_0).
# Outputs gpurun_out/bench_<name>.json; copy to profiles/rNN_bench_<name>.json.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
show() { python -c "
import json,sys; d=json.load(open(sys.argv[1])); c=d.get('cpu_baseline') or {}
print(sys.argv[2], d['value'], 'img/s', d['ms_per_step'], 'ms  p50', d['p50_latency_ms_batch1'], ' whole-forward TF', d['roofline']['whole_forward_tflops'], ' |dlogit|', c.get('max_abs_logit_diff_vs_gpu'), 'rel', c.get('rel_logit_diff_vs_gpu'))" $1 "$2"; }
timeout 900 python bench.py --model base --batch 1 --steps 100 --warmup 20 > gpurun_out/bench_base_b1.json 2> gpurun_out/bb.err; show gpurun_out/bench_base_b1.json "base b1"
timeout 900 python bench.py --model giant --dtype bf16 --batch 8 --steps 10 --warmup 3 > gpurun_out/bench_giant_bf16_b8.json 2> gpurun_out/bg.err; show gpurun_out/bench_giant_bf16_b8.json "giant bf16 b8"
timeout 900 python bench.py --wtype q8_0 --no-latency > gpurun_out/bench_large_q8_0.json 2> gpurun_out/bq8.err; show gpurun_out/bench_large_q8_0.json "large q8_0"
timeout 900 python bench.py --wtype q4_0 --no-latency > gpurun_out/bench_large_q4_0.json 2> gpurun_out/bq4.err; show gpurun_out/bench_large_q4_0.json "large q4_0"
