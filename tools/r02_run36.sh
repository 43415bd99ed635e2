#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SH="--shape qkv,qkv,1374,3072,1024 --shape ffn_in,gelu,1374,4096,1024"
{
for l in product max-ilp max-memory-clause product; do
  if [ $l = product ]; then unset DINOV2_HIP_LIB; else export DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_$l.so; fi
  echo "== $l"; DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
  timeout 300 python tools/kernel_bench.py --batch 1 --only attention 2>&1 | tail -1
  timeout 300 python tools/kernel_bench.py --iters 300 --ksplit $SH --shape ffn_out,resid,1374,1024,4096 2>&1 | tail -3
done
} > gpurun_out/run36.log 2>&1
cat gpurun_out/run36.log
