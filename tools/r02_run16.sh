#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "gemm" 2>&1 | tail -25
timeout 300 python tools/kernel_bench.py 2>&1 | tail -6
timeout 300 python tools/kernel_bench.py --shape p4k,plain,4096,4096,4096 --shape p8k,plain,8192,8192,8192 2>&1 | tail -2
} > gpurun_out/run16.log 2>&1
cat gpurun_out/run16.log
