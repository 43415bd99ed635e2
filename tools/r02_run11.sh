#!/bin/bash
# attention_kernel (version 1) timing-only ablations: which of MFMA / softmax VALU / LDS reads / staging / barrier the tile time is made of
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
echo "product v1:"; DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
for a in 2 4 6 24 30 32 65 97 127; do
  echo "ABL=$a v1:"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_a$a.so DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
done
for a in 30 32 65; do
  echo "ABL=$a v3:"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_a$a.so DINOV2_HIP_ATTN_V=3 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1
done
} > gpurun_out/run11.log 2>&1
cat gpurun_out/run11.log
