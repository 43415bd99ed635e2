#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
echo "== product (0.79 / 0.5)"; timeout 300 python tools/kernel_bench.py 2>&1 | head -4
echo "== forced 256-row tiles only"; DINOV2_HIP_GEMM_TILE=256 timeout 300 python tools/kernel_bench.py 2>&1 | head -4
for i in 1 2 3 4 5; do echo "== pl$i"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_pl$i.so timeout 300 python tools/kernel_bench.py 2>&1 | head -4; done
echo "== ViT-g b8 product"; timeout 300 python tools/kernel_bench.py --batch 8 --hidden 1536 --shape g_qkv,qkv,10992,4608,1536 --shape g_out,resid,10992,1536,1536 --shape g_in,swiglu,10992,8192,1536 --shape g_o2,resid,10992,1536,4096 2>&1 | tail -4
echo "== ViT-g b8 pl2"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_pl2.so timeout 300 python tools/kernel_bench.py --batch 8 --hidden 1536 --shape g_qkv,qkv,10992,4608,1536 --shape g_out,resid,10992,1536,1536 --shape g_in,swiglu,10992,8192,1536 --shape g_o2,resid,10992,1536,4096 2>&1 | tail -4
echo "== ViT-g b8 forced 256"; DINOV2_HIP_GEMM_TILE=256 timeout 300 python tools/kernel_bench.py --batch 8 --hidden 1536 --shape g_qkv,qkv,10992,4608,1536 --shape g_out,resid,10992,1536,1536 --shape g_in,swiglu,10992,8192,1536 --shape g_o2,resid,10992,1536,4096 2>&1 | tail -4
} > gpurun_out/run19.log 2>&1
cat gpurun_out/run19.log
