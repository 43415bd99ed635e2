#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for r in 1 2; do for c in 0_0 0_1 30_0 30_1; do
  echo "ABL_FAKE16=$c v1:"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_f$c.so DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
done; done
} > gpurun_out/run22.log 2>&1
cat gpurun_out/run22.log
