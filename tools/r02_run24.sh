#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for r in 1 2; do
echo "product v1:"; DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
for a in 1 2 3; do
  echo "STAGEPOS=$a v1:"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_sp$a.so DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
done; done
for a in 1 2 3; do DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_sp$a.so timeout 300 python -m pytest tests/test_gpu_ops.py -q -k attention 2>&1 | tail -1; done
} > gpurun_out/run24.log 2>&1
cat gpurun_out/run24.log
