#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
echo "product v1:"; DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
for a in 1 2 3 4; do
  for v in 1 3; do echo "PRIO=$a v$v:"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_p$a.so DINOV2_HIP_ATTN_V=$v timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1; done
done
echo "product v1:"; DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
} > gpurun_out/run15.log 2>&1
cat gpurun_out/run15.log
