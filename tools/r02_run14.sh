#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for a in 0 128 0 128; do
  echo "ABL=$a v1:"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_a$a.so DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
done
} > gpurun_out/run14.log 2>&1
cat gpurun_out/run14.log
