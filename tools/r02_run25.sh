#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SH="--shape qkv,qkv,1374,3072,1024 --shape ffn_in,gelu,1374,4096,1024 --shape b_qkv,qkv,1374,2304,768 --shape b_ffn_in,gelu,1374,3072,768 --shape q2,qkv,2748,3072,1024 --shape f2,gelu,2748,4096,1024"
{
echo "== product"; timeout 300 python tools/kernel_bench.py --iters 300 --ksplit $SH 2>&1 | tail -6
echo "== 128x128 KS=2"; DINO_TRY_K128=1 timeout 300 python tools/kernel_bench.py --iters 300 --ksplit $SH 2>&1 | tail -6
DINO_TRY_K128=1 timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "split_k" 2>&1 | tail -2
} > gpurun_out/run25.log 2>&1
cat gpurun_out/run25.log
