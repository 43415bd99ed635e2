#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
echo "product v1:"; DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
echo "v1, 8 waves:"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_w8.so DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
echo "v3, 2 waves:"; DINOV2_HIP_ATTN_V=3 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
echo "v3, 4 waves:"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_q4.so DINOV2_HIP_ATTN_V=3 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
} > gpurun_out/run12.log 2>&1
cat gpurun_out/run12.log
