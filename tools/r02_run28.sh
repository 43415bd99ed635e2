#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k attention 2>&1 | tail -1
for r in 1 2; do for l in before chunk_lb4 chunk_lb2; do
  echo "$l v1:"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_$l.so DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
done; done
echo "chunk_lb4 v3:"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_chunk_lb4.so DINOV2_HIP_ATTN_V=3 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
echo "before v3:"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_before.so DINOV2_HIP_ATTN_V=3 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
} > gpurun_out/run28.log 2>&1
cat gpurun_out/run28.log
