#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^$" | tail -15
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['p50_latency_ms_batch1'], d['roofline']['frac']); print({k:(v['avg_ms'], v.get('tflops')) for k,v in d['kernels'].items()})"
} > gpurun_out/run17.log 2>&1
cat gpurun_out/run17.log
