#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "gemm" 2>&1 | tail -2
for r in 1 2; do
echo "== new"; timeout 300 python tools/kernel_bench.py 2>&1 | head -5
echo "== prev"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_prev.so timeout 300 python tools/kernel_bench.py 2>&1 | head -5
done
} > gpurun_out/run20.log 2>&1
cat gpurun_out/run20.log
