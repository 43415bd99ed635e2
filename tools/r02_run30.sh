#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
echo "product v1:"; DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
for a in 8 16 10 18 26; do
  echo "ABL=$a v1:"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_a$a.so DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
done
} > gpurun_out/run30.log 2>&1
cat gpurun_out/run30.log
