#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for r in 1 2 3; do
echo "== product"; timeout 300 python tools/kernel_bench.py 2>&1 | head -4
echo "== no setprio"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_noprio.so timeout 300 python tools/kernel_bench.py 2>&1 | head -4
done
} > gpurun_out/run26.log 2>&1
cat gpurun_out/run26.log
