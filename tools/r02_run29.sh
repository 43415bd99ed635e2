#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for r in 1 2; do
echo "== product GM=8"; timeout 300 python tools/kernel_bench.py 2>&1 | head -4
for g in 4 16 32; do echo "== GM=$g"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_gm$g.so timeout 300 python tools/kernel_bench.py 2>&1 | head -4; done
done
} > gpurun_out/run29.log 2>&1
cat gpurun_out/run29.log
