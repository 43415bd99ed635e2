#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_alt.so timeout 600 python -m pytest tests/test_gpu_ops.py -q -k gemm 2>&1 | tail -1
for r in 1 2 3; do
echo "== product"; timeout 300 python tools/kernel_bench.py 2>&1 | head -4
echo "== alternating tile order"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_alt.so timeout 300 python tools/kernel_bench.py 2>&1 | head -4
done
} > gpurun_out/run32.log 2>&1
cat gpurun_out/run32.log
