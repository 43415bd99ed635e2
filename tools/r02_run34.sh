#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
echo "v4 product"; DINOV2_HIP_ATTN_V=4 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
for a in 1 2 4; do echo "v4 ABL=$a (1 no softmax, 2 no MFMA sections, 4 no staging)"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_a4_$a.so DINOV2_HIP_ATTN_V=4 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1; done
} > gpurun_out/run34.log 2>&1
cat gpurun_out/run34.log
